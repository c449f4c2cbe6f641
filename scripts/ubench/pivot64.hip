// Latency of the instructions the fp64 pivot chain is made of, as DEPENDENT chains (one wave, s_memtime = shader clocks):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pivot64 scripts/ubench/pivot64.hip && /tmp/pivot64
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 32
__device__ __forceinline__ double rl64(double v, int src) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int MODE>
__global__ void k(double* out, long long* t, double seed) {
    double x = seed + 1e-3 * threadIdx.x, y = seed * 0.5, z = 1.0;
    float xf = (float)x;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (MODE == 0) x = __builtin_fma(x, y, z);                       // dependent v_fma_f64
            else if (MODE == 1) x = __builtin_amdgcn_rsq(x) + 1.0;            // dependent v_rsq_f64 (+ add)
            else if (MODE == 2) { asm volatile("s_nop 0" : "+v"(x)); x = rl64(x, r & 31); x = x + z; }   // VALU -> 2 readlane -> SGPR -> VALU
            else if (MODE == 3) xf = __builtin_fmaf(xf, 0.5f, 1.0f);          // dependent v_fma_f32
            else if (MODE == 4) xf = __builtin_amdgcn_rsqf(xf) + 1.0f;        // dependent v_rsq_f32 (+ add)
            else if (MODE == 5) { asm volatile("s_nop 0" : "+v"(xf)); xf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xf), r & 31)) + 1.0f; }
            else if (MODE == 6) x = x * y;                                    // dependent v_mul_f64
            else if (MODE == 7) {                                             // the whole refinement: rsq + third-order step
                const double y0 = __builtin_amdgcn_rsq(x);
                const double e = __builtin_fma(-(x * y0), y0, 1.0);
                x = __builtin_fma(y0, e * __builtin_fma(0.375, e, 0.5), y0) + 1.0;
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x + xf;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
template <int MODE> void run(const char* nm) {
    double* o; long long* t; hipMalloc(&o, 512); hipMalloc(&t, 8);
    long long h = 0;
    for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, o, t, 1.25); hipDeviceSynchronize(); }
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%-64s %7.2f clocks per link\n", nm, (double)h / (256.0 * REP));
}
int main() {
    run<0>("v_fma_f64, dependent");
    run<6>("v_mul_f64, dependent");
    run<1>("v_rsq_f64 + v_add_f64, dependent");
    run<7>("rsq_f64 + third-order step + add (5 + 1 dependent ops)");
    run<2>("s_nop; 2 x v_readlane -> SGPR pair -> v_add_f64, dependent");
    run<3>("v_fma_f32, dependent");
    run<4>("v_rsq_f32 + v_add_f32, dependent");
    run<5>("s_nop; v_readlane -> SGPR -> v_add_f32, dependent");
    return 0;
}
