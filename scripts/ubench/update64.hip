// Issue cost of the fp64 diagonal block's rank-1 updates (tiles64.h pivot_phase64) WITHOUT the pivot chain: one wave, one matrix
// row per lane, 32 pivots, a[c] -= l * broadcast(l, lane c) for c > j -- 496 v_fma_f64 fed by 992 v_readlane_b32.
// Variants differ in how many broadcasts are in flight before their FMAs issue (s_memtime = shader clocks):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/update64 scripts/ubench/update64.hip && /tmp/update64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>

#define RL(s, lane) "v_readlane_b32 s" #s ", %[lo], " lane "\n\t"
#define RL2(s0, s1, lane) "v_readlane_b32 s" #s0 ", %[lo], " lane "\n\tv_readlane_b32 s" #s1 ", %[hi], " lane "\n\t"

__device__ __forceinline__ void g3(double& c0, double& c1, double& c2, double l, int lo, int hi, int l0, int l1, int l2) {
    asm volatile("v_readlane_b32 s90, %4, %6\n\tv_readlane_b32 s91, %5, %6\n\t"
                 "v_readlane_b32 s92, %4, %7\n\tv_readlane_b32 s93, %5, %7\n\t"
                 "v_readlane_b32 s94, %4, %8\n\tv_readlane_b32 s95, %5, %8\n\t"
                 "v_fma_f64 %0, -%3, s[90:91], %0\n\tv_fma_f64 %1, -%3, s[92:93], %1\n\tv_fma_f64 %2, -%3, s[94:95], %2"
                 : "+v"(c0), "+v"(c1), "+v"(c2)
                 : "v"(l), "v"(lo), "v"(hi), "i"(l0), "i"(l1), "i"(l2)
                 : "s90", "s91", "s92", "s93", "s94", "s95");
}
__device__ __forceinline__ void g1(double& c0, double l, int lo, int hi, int l0) {
    asm volatile("v_readlane_b32 s90, %2, %4\n\tv_readlane_b32 s91, %3, %4\n\ts_nop 1\n\t"
                 "v_fma_f64 %0, -%1, s[90:91], %0"
                 : "+v"(c0) : "v"(l), "v"(lo), "v"(hi), "i"(l0) : "s90", "s91");
}
// eight broadcasts in flight: 16 readlanes into s[80:95], then their 8 FMAs
__device__ __forceinline__ void g8(double* c, double l, int lo, int hi, int b) {
    asm volatile(RL2(80, 81, "%[b]") RL2(82, 83, "%[b]+1") RL2(84, 85, "%[b]+2") RL2(86, 87, "%[b]+3")
                 RL2(88, 89, "%[b]+4") RL2(90, 91, "%[b]+5") RL2(92, 93, "%[b]+6") RL2(94, 95, "%[b]+7")
                 "v_fma_f64 %0, -%[l], s[80:81], %0\n\tv_fma_f64 %1, -%[l], s[82:83], %1\n\t"
                 "v_fma_f64 %2, -%[l], s[84:85], %2\n\tv_fma_f64 %3, -%[l], s[86:87], %3\n\t"
                 "v_fma_f64 %4, -%[l], s[88:89], %4\n\tv_fma_f64 %5, -%[l], s[90:91], %5\n\t"
                 "v_fma_f64 %6, -%[l], s[92:93], %6\n\tv_fma_f64 %7, -%[l], s[94:95], %7"
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
                 : [l] "v"(l), [lo] "v"(lo), [hi] "v"(hi), [b] "i"(b)
                 : "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95");
}

template <int MODE, int J>
__device__ __forceinline__ void pivot(double (&a)[32]) {
    double lv = a[J];
    asm volatile("s_nop 0" : "+v"(lv));
    const unsigned long long lu = __builtin_bit_cast(unsigned long long, lv);
    const int lo = (int)(unsigned)lu, hi = (int)(unsigned)(lu >> 32);
    int c = J + 1;
    if (MODE == 0) {
#pragma unroll
        for (; c + 2 < 32; c += 3) g3(a[c], a[c + 1], a[c + 2], lv, lo, hi, c, c + 1, c + 2);
#pragma unroll
        for (; c < 32; ++c) g1(a[c], lv, lo, hi, c);
    } else if (MODE == 1) {
#pragma unroll
        for (; c + 7 < 32; c += 8) g8(&a[c], lv, lo, hi, c);
#pragma unroll
        for (; c + 2 < 32; c += 3) g3(a[c], a[c + 1], a[c + 2], lv, lo, hi, c, c + 1, c + 2);
#pragma unroll
        for (; c < 32; ++c) g1(a[c], lv, lo, hi, c);
    } else if (MODE == 2) {                 // readlanes only (same count), results folded in with cheap s_ ops afterwards
#pragma unroll
        for (; c < 32; ++c) asm volatile("v_readlane_b32 s90, %0, %2\n\tv_readlane_b32 s91, %1, %2" : : "v"(lo), "v"(hi), "i"(c) : "s90", "s91");
    } else if (MODE == 3) {                 // FMAs only, against one SGPR pair read once
        asm volatile("v_readlane_b32 s90, %0, 1\n\tv_readlane_b32 s91, %1, 1\n\ts_nop 1" : : "v"(lo), "v"(hi) : "s90", "s91");
#pragma unroll
        for (; c < 32; ++c) asm volatile("v_fma_f64 %0, -%1, s[90:91], %0" : "+v"(a[c]) : "v"(lv) : "s90", "s91");
    } else if (MODE == 4) {                 // left to the compiler: builtin readlane + fma
#pragma unroll
        for (; c < 32; ++c) {
            const unsigned blo = (unsigned)__builtin_amdgcn_readlane(lo, c), bhi = (unsigned)__builtin_amdgcn_readlane(hi, c);
            a[c] = __builtin_fma(-lv, __builtin_bit_cast(double, ((unsigned long long)bhi << 32) | blo), a[c]);
        }
    }
}
template <int MODE, int... J>
__device__ __forceinline__ void all(double (&a)[32], std::integer_sequence<int, J...>) { (pivot<MODE, J>(a), ...); }

template <int MODE>
__global__ void k(double* out, long long* t, double seed) {
    double a[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) a[c] = seed + 1e-3 * (threadIdx.x + c);
    long long r0 = wall_clock64();
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) all<MODE>(a, std::make_integer_sequence<int, 32>());
    long long t1 = __builtin_amdgcn_s_memtime();
    long long r1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int c = 0; c < 32; ++c) s += a[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = r1 - r0; }
}
template <int MODE> void run(const char* nm) {
    double* o; long long* t; (void)hipMalloc(&o, 512); (void)hipMalloc(&t, 16);
    long long h[2] = {0, 0};
    for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, o, t, 1e-3); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("%-72s %8.0f clocks per 32 pivots  = %5.2f us by the 100 MHz wall clock (s_memtime ran at %4.0f MHz)\n", nm, (double)h[0] / 64.0, (double)h[1] / 64.0 * 0.01, (double)h[0] / ((double)h[1] * 0.01));
}
int main() {
    run<0>("groups of 3 broadcasts (tiles64.h today)");
    run<1>("groups of 8 broadcasts, remainder by 3 / 1");
    run<2>("the 992 v_readlane_b32 alone");
    run<3>("the 496 v_fma_f64 alone");
    run<4>("builtin readlane + fma, scheduled by the compiler");
    return 0;
}
