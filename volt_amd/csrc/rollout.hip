// Sequential posterior rollouts, bordered-Cholesky engine (7 hard part 3).
//   reference: voltron/rollout_utils.py:57-93 calling GeneratePrediction (:6-53) H times; every call
//   re-fills and re-factors S matrices of size (N+idx)^2 from scratch.
//
// Structure used (and only this): the train block of every sample's matrix is the same
// (train_stack_vol = log_vol_path.repeat(S,1), :72) and the cross block between the train points and
// the appended points is sample-independent.  So per series the host factors K_NN once (HIP potrf)
// and solves K_NN x = u once (volt_amd/rollout_engine.py: in closed form, or on the fp64
// factorisation); with rho = u'x (= q'q, q = L^-1 u) and tau = x'r_tr (= q'z_tr) every sample owns a small dense
// bordered problem of dimension idx <= H:
//     S_s   = C_s - rho 11'            (Schur complement of the appended points),  L_s = chol(S_s)
//     w_s   = L_s^-1 (k*_s - rho 1),   z_s = L_s^-1 (r_s - tau 1)
//     mean  = tau + w_s'z_s + m(x*),   var = k** - rho - w_s'w_s
// The new row of L_s at step idx+1 is w_s of step idx (the new column of K_tr is the previous k*),
// which is ordinary row-by-row Cholesky.  w_s itself is append-only for this kernel (see rollout_bordered_kernel);
// the full per-step re-substitution against the stored dense rows is kept as a cross-check mode.
//
// One wave per sample (a sample's recursion is sequential in idx and in the substitution index);
// 4 samples per workgroup, EWMA histories in LDS.  H <= 1024: a lane owns the entries 256 c + 4 lane + t of
// ceil(H / 256) chunks, so a stored row is read and written with 16-byte accesses, 1 KB per wave-instruction.
#include "common.h"
#include "../../include/volt_hip.h"

namespace volt {

__device__ __forceinline__ double wave_sum_d(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}


// Mean of the appended point idx for the EWMA family (EWMA.py:20-37, :39-54, :74-91, :94-113, :116-135) from the
// per-sample histories in LDS (k train-tail values followed by the values appended so far).  ma1 = plain EMA at
// the new index, e2new = EMA(EMA) there (dewma / tewma).
__device__ __forceinline__ float family_mean(int mode, int idx, int k, int lane, const float* sw, const float* hy,
                                             const float* he1, const float* he2, float ema_prev, float mr_theta,
                                             float mr_latent, float& ma1, float& e2new) {
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int j = lane; j < k; j += 64) {
        const double wj = (double)sw[j];
        a1 += wj * (double)hy[idx + j];
        if (mode == 1 || mode == 2) a2 += wj * (double)he1[idx + j];
        if (mode == 2) a3 += wj * (double)he2[idx + j];
    }
    ma1 = (float)wave_sum_d(a1);
    float mstar = ma1;
    e2new = 0.f;
    if (mode == 1 || mode == 2) {
        e2new = (float)wave_sum_d(a2);                       // EMA(EMA) at the new index
        if (mode == 1) mstar = 2.f * ma1 - e2new;
        else mstar = 3.f * ma1 - 3.f * e2new + (float)wave_sum_d(a3);
    } else if (mode == 3) {
        mstar = ma1 - mr_theta * (ema_prev - mr_latent);
    }
    return mstar;
}

struct RolloutParams {
    // per series g (G series), all device pointers
    const double* rho;       // [G]   q'q     (fp64: it is subtracted from the CumTrapz sums, see `base` below)
    const double* tau;       // [G]   q'z_tr
    const double* acc0;      // [G]   fp64 running CumTrapz sum through train point N-1 (full weights)
    const float* dx;         // [G]   x[1]-x[0]
    const float* hist_y;     // [G,k] padded train series tail  Y[N-k .. N-1]
    const float* hist_e1;    // [G,k] EMA tail    ema[N-k .. N-1]     (dewma / tewma)
    const float* hist_e2;    // [G,k] EMA(EMA) tail                    (tewma)
    const float* ema_prev;   // [G]   plain ema[N-1]                    (meanrevert)
    const float* mr_latent;  // [G]   latent mean of the mean module    (meanrevert)
    const float* latent;     // [G]   Rollouts' latent_mean (theta != NULL) or unused
    const float* w;          // [k]   EWMA taps
    const float* pred_vol;   // [G,S,H]
    const float* z;          // [G,S,H]
    float* samples;          // [G,S,H]
    float* Ls;               // [G,S,rollout_sample_floats(H)] scratch: packed strictly-lower rows of the per-sample factor
    int* info;               // [G,S] 0; +step (1-based) of the first non-positive pivot of L_s (a local jitter was
                             //       applied); -step if the predictive variance stayed <= 0 after the jitter ladder
    int G, S, H, k;
    int mean_mode;           // 0 ewma, 1 dewma, 2 tewma, 3 meanrevert, 4 given: hist_e1 = the mean at the H appended points [G,H]
    int use_theta;
    float theta, mr_theta, jitter;
};

// The strictly-lower rows of a sample's factor are stored PACKED: row a (a entries; its diagonal lives in registers)
// takes ceil(a/4) 16-byte slots right behind row a-1.  With one row per KiB (round 2's first layout) every row read
// fetched whole 128-byte lines of its own -- 1.18x the algorithmic bytes (PMC: 8.25e9 read requests, all 128-B) --
// and the store was twice as large.  row_off(a) = 4 * sum_{m<a} ceil(m/4) floats.
__device__ __forceinline__ int row_off(int a) {
    const int m = a - 1;                       // sum_{i=1}^{m} ceil(i/4) with m = 4 q + r:  (q+1)(2q + r)
    if (m <= 0) return 0;
    const int q = m >> 2, r = m & 3;
    return 4 * (q + 1) * (2 * q + r);
}
__host__ __device__ inline size_t rollout_sample_floats(int H) {
    const int m = H - 1, q = m >> 2, r = m & 3;
    return m <= 0 ? 4 : (size_t)4 * (q + 1) * (2 * q + r) + 4;    // rows 1 .. H-1 (+ one slot of slack)
}

// NC = 256-entry chunks per factor row (H <= 256 NC): a lane owns the entries 256 c + 4 lane + t, so a stored row is
// read and written with NC 16-byte accesses per lane, each wave-instruction covering 1 KB of contiguous memory.
//
// Two ways to get w_s at step idx:
//   RESUB = false  APPEND-ONLY (default).  For the volatility kernel the right-hand side prefix is step-invariant -- the
//      covariance between appended point a and ANY later point is U_s[a] (k(x*, x_a) = V[min(a, *)]) -- and the stored
//      rows never change, so w_s(idx) = [w_s(idx-1), new entry]: one dot product against the row appended last (it is
//      still in registers), nothing is re-read, no scratch.  O(idx) flops per step, O(H^2) per path.
//   RESUB = true   FULL RE-SUBSTITUTION against the stored rows at every step -- what a kernel without that invariance would
//      need.  O(idx^2) words streamed per step (H^3/6 * 4 B per path, HBM-bound).  Same arithmetic in the same order, so
//      the two produce BITWISE identical paths (tested); kept as the cross-check and as the measured cost of not
//      using the invariance (bench.py reports it as redundant bytes).
template <int NC, bool RESUB>
__global__ __launch_bounds__(256) void rollout_bordered_kernel(RolloutParams p) {
    // Both instantiations must round identically (the test holds them to bitwise equality), so nothing here is left
    // to the compiler's choice of what to fuse: contraction off, every multiply-add that should be one is written as one.
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x * 4 + wave, g = blockIdx.y;
    const int H = p.H, k = p.k;
    const int hl = k + H;                                   // history length per level
    float* hy = lds + (size_t)wave * 3 * hl;
    float* he1 = hy + hl;
    float* he2 = he1 + hl;
    float* sw = lds + (size_t)4 * 3 * hl;                   // taps, shared by the 4 waves
    for (int j = threadIdx.x; j < k; j += 256) sw[j] = p.w[j];
    if (s < p.S) {
        for (int j = lane; j < k; j += 64) {
            hy[j] = p.hist_y[(size_t)g * k + j];
            he1[j] = (p.mean_mode == 1 || p.mean_mode == 2) ? p.hist_e1[(size_t)g * k + j] : 0.f;
            he2[j] = (p.mean_mode == 2) ? p.hist_e2[(size_t)g * k + j] : 0.f;
        }
    }
    __syncthreads();
    if (s >= p.S) return;

    const float tau = (float)p.tau[g], dx = p.dx[g], hdx = dx * 0.5f;
    const size_t row = ((size_t)g * p.S + s) * H;
    const float* pv = p.pred_vol + row;
    const float* zz = p.z + row;
    float* out = p.samples + row;
    float* Ls = RESUB ? p.Ls + ((size_t)g * p.S + s) * rollout_sample_floats(H) : nullptr;
    // base = U_s[N+a] - rho, carried in fp64: the entries of the Schur complement S_s = C_s - rho 11' are
    // ~ dx vol^2 (1e-4 .. 1e-9) on top of rho ~ V[N-1] ~ 1, so forming U_s and rho in fp32 first loses them
    // (62 of 80,000 paths at N = 4096 lost a pivot that way); the differences themselves are fine in fp32.
    double base = p.acc0[g] - p.rho[g];
    float ema_prev = (p.mean_mode == 3) ? p.ema_prev[g] : 0.f;
    int bad = 0;

    // entry (c, t) of these arrays belongs to appended point b = 256 c + 4 lane + t
    float U[NC][4], rd[NC][4], zs[NC][4], wraw[NC][4];      // U_s - rho;  1 / L_s[b][b];  z_s[b];  (U_s - rho) - sum (un-normalised w)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) U[c][t] = rd[c][t] = zs[c][t] = wraw[c][t] = 0.f;

    for (int idx = 0; idx < H; ++idx) {
        float wv[NC][4];
        if constexpr (RESUB) {
            // ---- w_s = L_s^-1 (U_s - rho): row-oriented forward substitution against ALL stored rows ----
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t) wv[c][t] = U[c][t];
            // Rows are streamed in groups of 4 with the next group's loads issued before the current group is
            // consumed: the substitution is a dependent chain (row a needs w[a-1]) but the addresses are not, so 8
            // rows per wave stay in flight and the stream runs at memory bandwidth, not one latency per row.
            float cur[4][NC][4], nxt[4][NC][4];
            auto load_rows = [&](float (&dst)[4][NC][4], int a0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a = a0 + r;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const int b0 = 256 * c + 4 * lane;
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if (a < idx && b0 < a) v = *reinterpret_cast<const f32x4*>(Ls + row_off(a) + b0);
#pragma unroll
                        for (int t = 0; t < 4; ++t) dst[r][c][t] = (b0 + t < a) ? v[t] : 0.f;   // beyond b < a: never written
                    }
                }
            };
            load_rows(cur, 1);                               // row 0 has no off-diagonal part
            for (int a0 = 1; a0 < idx; a0 += 4) {
                load_rows(nxt, a0 + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a = a0 + r;
                    if (a < idx) {                           // wave-uniform
                        float part = 0.f;
#pragma unroll
                        for (int c = 0; c < NC; ++c)
#pragma unroll
                            for (int t = 0; t < 4; ++t)                                            // zero beyond b < a
                                part = __fmaf_rn(cur[r][c][t], __fmul_rn(wv[c][t], rd[c][t]), part);
                        const float dot = wave_sum_f(part);
#pragma unroll
                        for (int c = 0; c < NC; ++c)
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                if (256 * c + 4 * lane + t == a) wv[c][t] -= dot;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < NC; ++c)
#pragma unroll
                        for (int t = 0; t < 4; ++t) cur[r][c][t] = nxt[r][c][t];
            }
        } else {
            // ---- append-only: entries b < idx-1 stand; the new one, b = idx-1, is the same dot product the full
            // substitution would take against row idx-1 -- whose entries are the w_s of the step before, in registers
            if (idx >= 2) {
                const int a = idx - 1;
                float part = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float wfin = __fmul_rn(wraw[c][t], rd[c][t]);
                        part = __fmaf_rn((256 * c + 4 * lane + t < a) ? wfin : 0.f, wfin, part);
                    }
                const float dot = wave_sum_f(part);
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (256 * c + 4 * lane + t == a) wraw[c][t] -= dot;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t) wv[c][t] = wraw[c][t];
        }
        float ww = 0.f, wz = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int b = 256 * c + 4 * lane + t;
                wv[c][t] = (b < idx) ? __fmul_rn(wv[c][t], rd[c][t]) : 0.f;    // now wv = w_s
                ww = __fmaf_rn(wv[c][t], wv[c][t], ww);
                wz = __fmaf_rn(wv[c][t], zs[c][t], wz);
            }
        ww = wave_sum_f(ww);
        wz = wave_sum_f(wz);

        // ---- mean of the new point: EWMA family on the stacked series (EWMA.py:20-37) ----------
        // (mode 4: a mean that is a function of x alone -- constant / linear / log-linear, the weather driver's default,
        // experiments/weather/GPGenerator.py:68-82 -- is history-free: the host evaluated it at the test points)
        float ma1 = 0.f, e2new = 0.f;
        const float mstar = (p.mean_mode == 4) ? p.hist_e1[(size_t)g * H + idx]
                                               : family_mean(p.mean_mode, idx, k, lane, sw, hy, he1, he2, ema_prev, p.mr_theta,
                                                             (p.mean_mode == 3) ? p.mr_latent[g] : 0.f, ma1, e2new);

        // ---- conditional and draw (rollout_utils.py:36-53) --------------------------------------
        const float v = pv[idx];
        const float v2 = v * v;
        const float kss = (float)(base + (double)__fmul_rn(hdx, v2));      // k** - rho; last CumTrapz weight halved
        float pm = tau + wz + mstar;
        if (p.use_theta) pm = __fmaf_rn(-p.theta, pm - p.latent[g], pm);
        float pvar = kss - ww;
        if (!(pvar > 0.f)) {                                 // psd_safe_cholesky(pred_cov, jitter) ladder
            float jit = p.jitter;
            int tries = 0;
            while (!(pvar + jit > 0.f) && tries < 2) { jit *= 10.f; ++tries; }
            if (pvar + jit > 0.f) pvar += jit;
            else { if (bad >= 0) bad = -(idx + 1); pvar = 0.f; }     // ladder exhausted: the reference raises NotPSDError
        }
        const float smp = __fmaf_rn(sqrtf(pvar), zz[idx], pm);
        if (lane == 0) out[idx] = smp;

        // ---- append the point to the conditioning set --------------------------------------------
        base += (double)__fmul_rn(dx, v2);                   // full weight from now on
        const float Unew = (float)base;
        float d2 = Unew - ww;                                // next pivot of L_s
        if (!(d2 > 0.f)) {
            if (!bad) bad = idx + 1;
            d2 = fmaxf(p.jitter, 1e-12f);
        }
        const float ell = sqrtf(d2), rell = 1.f / ell;
        const float znew = ((smp - mstar) - tau - wz) * rell;
        // row idx of L_s = [w_s, ell]: the off-diagonal part goes to the packed store (re-substitution mode only),
        // ell stays in registers (rd)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int b0 = 256 * c + 4 * lane;
            f32x4 rowv;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int b = b0 + t;
                rowv[t] = (b < idx) ? wv[c][t] : ((b == idx) ? ell : 0.f);
                if (b == idx) {
                    U[c][t] = Unew;
                    wraw[c][t] = Unew;                       // append-only: (U_s - rho) before its own row's dot product
                    rd[c][t] = rell;
                    zs[c][t] = znew;
                }
            }
            if (RESUB && b0 < idx) *reinterpret_cast<f32x4*>(Ls + row_off(idx) + b0) = rowv;
        }
        if (lane == 0) {
            hy[k + idx] = smp;
            he1[k + idx] = ma1;
            he2[k + idx] = e2new;
        }
        ema_prev = ma1;
        __builtin_amdgcn_wave_barrier();
        // make this wave's own LDS/global writes visible to its later reads
        __threadfence_block();
    }
    if (lane == 0) p.info[(size_t)g * p.S + s] = bad;
}


// ---------------------------------------------------------------------------------------------------------
// Rollouts of a GP whose kernel does not depend on the sample (voltron/rollout_utils.py:95-115, nonvol_rollouts):
// every sample shares one factorisation of the (N+H)^2 matrix, and the new entry of L^-1 r for an appended draw
// is gamma_j z_j whatever the history, so the GP part of every path is e = c + M z (one small GEMM on the host
// side, volt_amd/rollout_engine.py).  What stays sequential per sample is the moving-average mean of the
// stacked series: samples[idx] = mean_s(idx) + e[idx], with mean_s over the sample's own earlier draws.
struct SharedParams {
    const float *hist_y, *hist_e1, *hist_e2, *ema_prev, *mr_latent, *w, *e;
    float* samples;
    int G, S, H, k, mean_mode;
    float mr_theta;
};

__global__ __launch_bounds__(256) void rollout_shared_kernel(SharedParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x * 4 + wave, g = blockIdx.y;
    const int H = p.H, k = p.k, hl = k + H;
    float* hy = lds + (size_t)wave * 3 * hl;
    float* he1 = hy + hl;
    float* he2 = he1 + hl;
    float* sw = lds + (size_t)4 * 3 * hl;
    for (int j = threadIdx.x; j < k; j += 256) sw[j] = p.w[j];
    if (s < p.S) {
        for (int j = lane; j < k; j += 64) {
            hy[j] = p.hist_y[(size_t)g * k + j];
            he1[j] = (p.mean_mode == 1 || p.mean_mode == 2) ? p.hist_e1[(size_t)g * k + j] : 0.f;
            he2[j] = (p.mean_mode == 2) ? p.hist_e2[(size_t)g * k + j] : 0.f;
        }
    }
    __syncthreads();
    if (s >= p.S) return;
    const size_t row = ((size_t)g * p.S + s) * H;
    float ema_prev = (p.mean_mode == 3) ? p.ema_prev[g] : 0.f;
    for (int idx = 0; idx < H; ++idx) {
        float ma1, e2new;
        const float mstar = family_mean(p.mean_mode, idx, k, lane, sw, hy, he1, he2, ema_prev, p.mr_theta,
                                        (p.mean_mode == 3) ? p.mr_latent[g] : 0.f, ma1, e2new);
        const float smp = mstar + p.e[row + idx];
        if (lane == 0) {
            p.samples[row + idx] = smp;
            hy[k + idx] = smp;
            he1[k + idx] = ma1;
            he2[k + idx] = e2new;
        }
        ema_prev = ma1;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
}

}  // namespace volt

extern "C" {

size_t volt_rollout_scratch_bytes(int G, int S, int H) {
    if (G <= 0 || S <= 0 || H <= 0) return 0;
    return (size_t)G * S * volt::rollout_sample_floats(H) * sizeof(float);      // packed strictly-lower rows, 16-byte slots
}

int volt_rollout_bordered_f32(const double* rho, const double* tau, const double* acc0, const float* dx,
                              const float* hist_y, const float* hist_e1, const float* hist_e2, const float* ema_prev,
                              const float* mr_latent, const float* latent, const float* w, const float* pred_vol,
                              const float* z, float* samples, float* scratch, int* info, int G, int S, int H, int k,
                              int mean_mode, int use_theta, float theta, float mr_theta, float jitter, void* stream) {
    using namespace volt;
    if (!rho) return -1;
    if (!tau) return -2;
    if (!acc0) return -3;
    if (!dx) return -4;
    if (!hist_y) return -5;
    if (!w) return -11;
    if (!pred_vol) return -12;
    if (!z) return -13;
    if (!samples) return -14;
    if (!info) return -16;
    if (G < 0) return -17;
    if (S < 0) return -18;
    if (H < 1 || H > VOLT_ROLLOUT_MAX_H) return -19;
    if (k < 1 || k > 2048) return -20;
    if (mean_mode < 0 || mean_mode > 4) return -21;
    if ((mean_mode == 1 || mean_mode == 2 || mean_mode == 4) && !hist_e1) return -6;
    if (mean_mode == 2 && !hist_e2) return -7;
    if (mean_mode == 3 && (!ema_prev || !mr_latent)) return -8;
    if (use_theta && !latent) return -10;
    if (G == 0 || S == 0) return 0;
    RolloutParams p{rho, tau, acc0, dx, hist_y, hist_e1, hist_e2, ema_prev, mr_latent, latent, w, pred_vol, z,
                    samples, scratch, info, G, S, H, k, mean_mode, use_theta, theta, mr_theta, jitter};
    const size_t lds = ((size_t)4 * 3 * (k + H) + k) * sizeof(float);
    if (lds > 160 * 1024) return -20;
    const dim3 grid((S + 3) / 4, G);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
#define VOLT_ROLLOUT_LAUNCH1(NC, RS)                                                                                   \
    do {                                                                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(rollout_bordered_kernel<NC, RS>),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
        if (e != hipSuccess) return (int)e;                                                                            \
        hipLaunchKernelGGL((rollout_bordered_kernel<NC, RS>), grid, dim3(256), lds, s, p);                           \
    } while (0)
#define VOLT_ROLLOUT_LAUNCH(NC)                                                                                        \
    do {                                                                                                               \
        if (scratch) VOLT_ROLLOUT_LAUNCH1(NC, true);                                                                   \
        else VOLT_ROLLOUT_LAUNCH1(NC, false);                                                                          \
    } while (0)
    if (H <= 256) VOLT_ROLLOUT_LAUNCH(1);
    else if (H <= 512) VOLT_ROLLOUT_LAUNCH(2);
    else VOLT_ROLLOUT_LAUNCH(4);
#undef VOLT_ROLLOUT_LAUNCH
#undef VOLT_ROLLOUT_LAUNCH1
    VOLT_LAUNCH_CHECK();
    return 0;
}

int volt_rollout_shared_f32(const float* hist_y, const float* hist_e1, const float* hist_e2, const float* ema_prev,
                            const float* mr_latent, const float* w, const float* e, float* samples, int G, int S, int H,
                            int k, int mean_mode, float mr_theta, void* stream) {
    using namespace volt;
    if (!hist_y) return -1;
    if (!w) return -6;
    if (!e) return -7;
    if (!samples) return -8;
    if (G < 0) return -9;
    if (S < 0) return -10;
    if (H < 1 || H > 4096) return -11;
    if (k < 1 || k > 2048) return -12;
    if (mean_mode < 0 || mean_mode > 3) return -13;
    if ((mean_mode == 1 || mean_mode == 2) && !hist_e1) return -2;
    if (mean_mode == 2 && !hist_e2) return -3;
    if (mean_mode == 3 && (!ema_prev || !mr_latent)) return -4;
    if (G == 0 || S == 0) return 0;
    SharedParams p{hist_y, hist_e1, hist_e2, ema_prev, mr_latent, w, e, samples, G, S, H, k, mean_mode, mr_theta};
    const size_t lds = ((size_t)4 * 3 * (k + H) + k) * sizeof(float);
    if (lds > 160 * 1024) return -11;
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(rollout_shared_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return (int)err;
    hipLaunchKernelGGL(rollout_shared_kernel, dim3((S + 3) / 4, G), dim3(256), lds, (hipStream_t)stream, p);
    VOLT_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
