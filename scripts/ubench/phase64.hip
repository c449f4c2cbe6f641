// The fp64 diagonal block's pivot phase (tiles64.h pivot_phase64: one wave, one row per lane, 32 pivots) taken apart:
// the whole phase against an LDS image, then with one ingredient removed at a time (s_memtime = shader clocks, 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/phase64 scripts/ubench/phase64.hip && /tmp/phase64
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int TS = 128, DT64 = TS + 1;
enum { NO_RDIAG = 1, NO_BAD = 2, NO_LDS = 4, NO_UPDATE = 8, ONE_NEWTON = 16, NO_CHAIN = 32, WAVES3 = 64, RECORD = 128, DPP = 256, PIPE = 512, WEAVE = 1024 };

__device__ __forceinline__ double rl64(double v, int src) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void rl_fma64_3(double& c0, double& c1, double& c2, double l, int lo, int hi, int l0, int l1, int l2) {
    asm volatile("v_readlane_b32 s90, %4, %6\n\tv_readlane_b32 s91, %5, %6\n\t"
                 "v_readlane_b32 s92, %4, %7\n\tv_readlane_b32 s93, %5, %7\n\t"
                 "v_readlane_b32 s94, %4, %8\n\tv_readlane_b32 s95, %5, %8\n\t"
                 "v_fma_f64 %0, -%3, s[90:91], %0\n\tv_fma_f64 %1, -%3, s[92:93], %1\n\tv_fma_f64 %2, -%3, s[94:95], %2"
                 : "+v"(c0), "+v"(c1), "+v"(c2)
                 : "v"(l), "v"(lo), "v"(hi), "i"(l0), "i"(l1), "i"(l2)
                 : "s90", "s91", "s92", "s93", "s94", "s95");
}
__device__ __forceinline__ void rl_fma64_2(double& c0, double& c1, double l, int lo, int hi, int l0, int l1) {
    asm volatile("v_readlane_b32 s90, %3, %5\n\tv_readlane_b32 s91, %4, %5\n\t"
                 "v_readlane_b32 s92, %3, %6\n\tv_readlane_b32 s93, %4, %6\n\t"
                 "v_fma_f64 %0, -%2, s[90:91], %0\n\ts_nop 0\n\tv_fma_f64 %1, -%2, s[92:93], %1"
                 : "+v"(c0), "+v"(c1)
                 : "v"(l), "v"(lo), "v"(hi), "i"(l0), "i"(l1)
                 : "s90", "s91", "s92", "s93");
}
__device__ __forceinline__ void rl_fma64_1(double& c0, double l, int lo, int hi, int l0) {
    asm volatile("v_readlane_b32 s90, %2, %4\n\tv_readlane_b32 s91, %3, %4\n\ts_nop 1\n\t"
                 "v_fma_f64 %0, -%1, s[90:91], %0"
                 : "+v"(c0) : "v"(l), "v"(lo), "v"(hi), "i"(l0) : "s90", "s91");
}

template <int F>
__device__ __forceinline__ void pivot_phase64(double* __restrict__ sT, double* __restrict__ rdiag, int kb, int prow, bool own, int& bad) {
    const int lane = threadIdx.x & 63, l31 = lane & 31;
    const bool up = lane >= 32;
    double* rowp = up ? sT + (32 * (prow < 0 ? kb : prow) + l31) * DT64 + 32 * kb : sT + (32 * kb + l31) * DT64 + 32 * kb;
    const bool live = !up || prow >= 0;
    double a[32], rsv = 1.0;
#pragma unroll
    for (int c = 0; c < 32; ++c) a[c] = (F & NO_LDS) ? (c == l31 && !up ? 40.0 : 0.01 * (c + l31)) : (live ? rowp[c] : 0.0);
    double rep0p = 0.0, rep1p = 0.0, nlp = 0.0;                      // PIPE: pivot j-1's replicated l column and -l
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        double l;
        if (F & NO_CHAIN) {
            l = a[j];
        } else if (F & WEAVE) {
            // PIPE with the order pinned: pivot j-1's columns j+2.. are dealt out between the dependent instructions of pivot j's chain
            const int n = 30 - j, per = (n + 6) / 7;
#define VOLT_SB __builtin_amdgcn_sched_barrier(0)
#define VOLT_SLOT(s)                                                                                                              \
    VOLT_SB;                                                                                                                      \
    if (j >= 1) {                                                                                                                 \
        _Pragma("unroll") for (int c = j + 2 + (s) * per; c < j + 2 + ((s) + 1) * per && c < 32; ++c) {                           \
            if (c < 16) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(rep0p), "v"(nlp), "i"(c & 15)); \
            else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(rep1p), "v"(nlp), "i"(c & 15));       \
        }                                                                                                                         \
    }                                                                                                                             \
    VOLT_SB;
            const double d = rl64(a[j], j);
            double y = __builtin_amdgcn_rsq(d);
            const double h = 0.5 * d;
            VOLT_SLOT(0)
            double t = h * y;
            VOLT_SLOT(1)
            double e = __builtin_fma(-y, t, 1.5);
            VOLT_SLOT(2)
            y = y * e;
            VOLT_SLOT(3)
            t = h * y;
            VOLT_SLOT(4)
            e = __builtin_fma(-y, t, 1.5);
            VOLT_SLOT(5)
            const double rs = y * e;
            VOLT_SLOT(6)
            l = a[j] * rs;
            a[j] = l;
            rsv = (lane == j) ? rs : rsv;
            double lv = l;
            asm volatile("s_nop 0" : "+v"(lv));
            const unsigned long long lu = __builtin_bit_cast(unsigned long long, lv);
            const int llo = (int)(unsigned)lu, lhi = (int)(unsigned)(lu >> 32);
            double rep0 = 0.0, rep1 = 0.0;
            if (j + 2 < 16) {
                const unsigned r0 = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (lane & 15), llo), r1 = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (lane & 15), lhi);
                rep0 = __builtin_bit_cast(double, ((unsigned long long)r1 << 32) | r0);
            }
            if (j + 2 < 32) {
                const unsigned r0 = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (16 + (lane & 15)), llo), r1 = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (16 + (lane & 15)), lhi);
                rep1 = __builtin_bit_cast(double, ((unsigned long long)r1 << 32) | r0);
            }
            if (j >= 1 && j + 1 < 32) {
                if (j + 1 < 16) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a[j + 1]) : "v"(rep0p), "v"(nlp), "i"((j + 1) & 15));
                else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a[j + 1]) : "v"(rep1p), "v"(nlp), "i"((j + 1) & 15));
            }
            if (j < 31) rl_fma64_1(a[j + 1], lv, llo, lhi, j + 1);
            VOLT_SB;
            rep0p = rep0; rep1p = rep1; nlp = -l;
            continue;
        } else {
            const double d = rl64(a[j], j);
            if (!(F & (NO_BAD | RECORD))) if (!(d > 0.0) && bad == 0) bad = 32 * kb + j + 1;
            double rs = __builtin_amdgcn_rsq(d);
            rs = rs * (1.5 - 0.5 * d * rs * rs);
            if (!(F & ONE_NEWTON)) rs = rs * (1.5 - 0.5 * d * rs * rs);
            l = a[j] * rs;
            a[j] = l;
            if (!(F & (NO_RDIAG | RECORD))) if (own && lane == 0) rdiag[32 * kb + j] = rs;
            if (F & RECORD) rsv = (lane == j) ? rs : rsv;       // lane j keeps 1 / L[j][j]: stored, and checked, once after the loop
        }
        if (F & PIPE) {
            // the same instructions, one pivot apart: pivot j's ds_bpermute pair flies while pivot j-1's columns are updated
            double lv = l;
            asm volatile("s_nop 0" : "+v"(lv));
            const unsigned long long lu = __builtin_bit_cast(unsigned long long, lv);
            const int llo = (int)(unsigned)lu, lhi = (int)(unsigned)(lu >> 32);
            double rep0 = 0.0, rep1 = 0.0;
            if (j + 2 < 16) {
                const unsigned r0 = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (lane & 15), llo), r1 = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (lane & 15), lhi);
                rep0 = __builtin_bit_cast(double, ((unsigned long long)r1 << 32) | r0);
            }
            if (j + 2 < 32) {
                const unsigned r0 = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (16 + (lane & 15)), llo), r1 = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (16 + (lane & 15)), lhi);
                rep1 = __builtin_bit_cast(double, ((unsigned long long)r1 << 32) | r0);
            }
            if (j >= 1 && j + 1 < 32) {
                if (j + 1 < 16) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a[j + 1]) : "v"(rep0p), "v"(nlp), "i"((j + 1) & 15));
                else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a[j + 1]) : "v"(rep1p), "v"(nlp), "i"((j + 1) & 15));
            }
            if (j < 31) rl_fma64_1(a[j + 1], lv, llo, lhi, j + 1);
            if (j >= 1) {
#pragma unroll
                for (int c = j + 2; c < 32; ++c) {
                    if (c < 16) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(rep0p), "v"(nlp), "i"(c & 15));
                    else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(rep1p), "v"(nlp), "i"(c & 15));
                }
            }
            rep0p = rep0; rep1p = rep1; nlp = -l;
        } else if ((F & DPP) && j < 31) {
            // column j+1 (the next pivot's) straight away through an SGPR broadcast; the others from two copies of the l column --
            // lanes 0..15's and lanes 16..31's, each repeated in all four rows of 16 lanes (ds_bpermute) -- by v_fmac_f64 with a
            // row_newbcast DPP operand: one instruction per column instead of two v_readlane + one v_fma
            double lv = l;
            asm volatile("s_nop 0" : "+v"(lv));
            const unsigned long long lu = __builtin_bit_cast(unsigned long long, lv);
            const int llo = (int)(unsigned)lu, lhi = (int)(unsigned)(lu >> 32);
            rl_fma64_1(a[j + 1], lv, llo, lhi, j + 1);
            const double nl = -l;
            double rep0 = 0.0, rep1;
            if (j + 2 < 16) {
                const unsigned r0 = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (lane & 15), llo), r1 = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (lane & 15), lhi);
                rep0 = __builtin_bit_cast(double, ((unsigned long long)r1 << 32) | r0);
            }
            {
                const unsigned r0 = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (16 + (lane & 15)), llo), r1 = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (16 + (lane & 15)), lhi);
                rep1 = __builtin_bit_cast(double, ((unsigned long long)r1 << 32) | r0);
            }
#pragma unroll
            for (int c = j + 2; c < 32; ++c) {
                if (c < 16) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(rep0), "v"(nl), "i"(c & 15));
                else asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(rep1), "v"(nl), "i"(c & 15));
            }
        } else if (!(F & NO_UPDATE)) {
            double lv = l;
            asm volatile("s_nop 0" : "+v"(lv));
            const unsigned long long lu = __builtin_bit_cast(unsigned long long, lv);
            const int llo = (int)(unsigned)lu, lhi = (int)(unsigned)(lu >> 32);
            int c = j + 1;
#pragma unroll
            for (; c + 2 < 32; c += 3) rl_fma64_3(a[c], a[c + 1], a[c + 2], lv, llo, lhi, c, c + 1, c + 2);
            if (c + 1 < 32) rl_fma64_2(a[c], a[c + 1], lv, llo, lhi, c, c + 1);
            else if (c < 32) rl_fma64_1(a[c], lv, llo, lhi, c);
        }
    }
    if (F & RECORD) {
        // a pivot d that is not > 0 (or NaN) leaves rs NaN (rsq(0) = inf, times 0 in the Newton step; rsq(d < 0) = NaN), and only LATER pivots inherit it
        const unsigned nb = (unsigned)__builtin_amdgcn_ballot_w64(!(rsv > 0.0));
        if (nb != 0 && bad == 0) bad = 32 * kb + __builtin_ctz(nb) + 1;
        if (own && !up) rdiag[32 * kb + l31] = rsv;
    }
    if (F & NO_LDS) {
        double s = 0;
#pragma unroll
        for (int c = 0; c < 32; ++c) s += a[c];
        if (s == 123.456) rowp[0] = s;
    } else if (up) {
        if (prow >= 0) {
#pragma unroll
            for (int c = 0; c < 32; ++c) rowp[c] = a[c];
        }
    } else if (own) {
#pragma unroll
        for (int c = 0; c < 32; ++c) rowp[c] = (c <= l31) ? a[c] : 0.0;
    }
}

template <int F>
__global__ void __launch_bounds__(256) k(long long* t, int* out, double* chk) {
    extern __shared__ double sT[];
    double* rdiag = sT + TS * DT64;
    const int wave = threadIdx.x >> 6;
    long long sum = 0;
    int bad = 0;
#pragma unroll 1
    for (int it = 0; it < 32; ++it) {
        for (int e = threadIdx.x; e < TS * TS; e += blockDim.x) {
            const int r = e >> 7, c = e & 127;
            sT[r * DT64 + c] = (r == c) ? 40.0 + 0.1 * it : 0.01 + 1e-4 * ((r * 7 + c * 7) & 31);
        }
        __syncthreads();
        const long long t0 = __builtin_amdgcn_s_memtime();
        const int kb = 1, npw = (F & WAVES3) ? 2 : 1;
        if (wave < npw) pivot_phase64<F>(sT, rdiag, kb, kb + 1 + wave <= 3 ? kb + 1 + wave : -1, wave == 0, bad);
        __syncthreads();
        const long long t1 = __builtin_amdgcn_s_memtime();
        sum += t1 - t0;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        t[0] = sum; out[0] = bad;
        double s1 = 0, s2 = 0;                       // the factor the last pass left in the image: L_11, L_21 (and L_31 with two waves), 1 / diag
        for (int r = 32; r < 128; ++r) for (int c = 32; c < 64; ++c) { const double v = sT[r * DT64 + c]; s1 += v * (1 + 0.001 * ((r * 31 + c) % 17)); s2 += v * v; }
        for (int c = 32; c < 64; ++c) s1 += rdiag[c];
        chk[0] = s1; chk[1] = s2;
    }
}
template <int F> void run(const char* nm) {
    long long* t; int* o; double* c; (void)hipMalloc(&t, 8); (void)hipMalloc(&o, 4); (void)hipMalloc(&c, 16);
    double hc[2]; int hb = 0;
    long long h = 0;
    const size_t lds = (TS * DT64 + 160) * 8;
    (void)hipFuncSetAttribute((const void*)k<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL(k<F>, dim3(1), dim3(256), lds, 0, t, o, c); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(&hb, o, 4, hipMemcpyDeviceToHost);
    printf("%-64s %7.0f clocks = %5.2f us per 32 pivots   [bad %d, checks %.17g %.17g]\n", nm, (double)h / 32.0, (double)h / 32.0 / 2400.0, hb, hc[0], hc[1]);
}
int main() {
    run<0>("the phase as it is (one wave)");
    run<WAVES3>("the phase as it is (two waves side by side)");
    run<NO_RDIAG>("without the reciprocal-pivot store");
    run<NO_BAD>("without the d > 0 check");
    run<NO_RDIAG | NO_BAD>("without either");
    run<NO_LDS>("rows not loaded from / stored to the image");
    run<ONE_NEWTON>("one Newton step");
    run<NO_UPDATE>("no rank-1 updates (chains independent)");
    run<NO_CHAIN>("no pivot chain (updates only)");
    run<NO_CHAIN | NO_LDS>("updates only, no image");
    run<NO_RDIAG | NO_BAD | NO_LDS>("chain + updates in registers only");
    run<RECORD>("reciprocal pivots kept in lane j, stored and checked once");
    run<RECORD | WAVES3>("the same, two waves side by side");
    run<RECORD | DPP>("... and the updates by v_fmac_f64 row_newbcast");
    run<RECORD | DPP | WAVES3>("the same, two waves side by side");
    run<RECORD | DPP | NO_LDS>("the same, one wave, registers only");
    run<RECORD | PIPE>("... and pivot j-1's columns updated under pivot j's bpermute");
    run<RECORD | PIPE | WAVES3>("the same, two waves side by side");
    run<RECORD | PIPE | NO_LDS>("the same, one wave, registers only");
    run<RECORD | WEAVE>("... dealt out between the instructions of the chain");
    run<RECORD | WEAVE | WAVES3>("the same, two waves side by side");
    run<RECORD | WEAVE | NO_LDS>("the same, one wave, registers only");
    return 0;
}
