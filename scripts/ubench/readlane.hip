// Issue cost of the broadcast + FMA idioms the diagonal-block factorisation is made of (one wave, s_memtime).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/readlane scripts/ubench/readlane.hip && /tmp/readlane
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int MODE>
__global__ void k(float* out, long long* t, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, l = seed * 0.5f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5};
    float s0, s1, s2, s3, s4, s5;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (MODE == 0)          // 3 readlane + 3 fma (rl_fma3)
                asm volatile("v_readlane_b32 %3, %6, 1\n\tv_readlane_b32 %4, %6, 2\n\tv_readlane_b32 %5, %6, 3\n\t"
                             "v_fma_f32 %0, -%6, %3, %0\n\tv_fma_f32 %1, -%6, %4, %1\n\tv_fma_f32 %2, -%6, %5, %2"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "=&s"(s0), "=&s"(s1), "=&s"(s2) : "v"(l));
            else if (MODE == 1)     // 6 fma, VGPR operands only
                asm volatile("v_fma_f32 %0, -%6, %6, %0\n\tv_fma_f32 %1, -%6, %6, %1\n\tv_fma_f32 %2, -%6, %6, %2\n\t"
                             "v_fma_f32 %3, -%6, %6, %3\n\tv_fma_f32 %4, -%6, %6, %4\n\tv_fma_f32 %5, -%6, %6, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(l));
            else if (MODE == 2)     // 6 readlane
                asm volatile("v_readlane_b32 %0, %6, 1\n\tv_readlane_b32 %1, %6, 2\n\tv_readlane_b32 %2, %6, 3\n\t"
                             "v_readlane_b32 %3, %6, 4\n\tv_readlane_b32 %4, %6, 5\n\tv_readlane_b32 %5, %6, 6"
                             : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "=&s"(s4), "=&s"(s5) : "v"(l));
            else if (MODE == 3)     // 6 fma with an SGPR operand (no readlane)
                asm volatile("v_fma_f32 %0, -%6, %7, %0\n\tv_fma_f32 %1, -%6, %7, %1\n\tv_fma_f32 %2, -%6, %7, %2\n\t"
                             "v_fma_f32 %3, -%6, %7, %3\n\tv_fma_f32 %4, -%6, %7, %4\n\tv_fma_f32 %5, -%6, %7, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(l), "s"(seed));
            else if (MODE == 4)     // 6 readlane + 3 pk_fma (two SGPRs feed one packed FMA)
                asm volatile("v_readlane_b32 %3, %9, 1\n\tv_readlane_b32 %4, %9, 2\n\tv_readlane_b32 %5, %9, 3\n\t"
                             "v_readlane_b32 %6, %9, 4\n\tv_readlane_b32 %7, %9, 5\n\tv_readlane_b32 %8, %9, 6\n\t"
                             "v_pk_fma_f32 %0, %10, %0, %0\n\tv_pk_fma_f32 %1, %10, %1, %1\n\tv_pk_fma_f32 %2, %10, %2, %2"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "=&s"(s4), "=&s"(s5)
                             : "v"(l), "v"(p0));
            else if (MODE == 5)     // 6 DPP row_bcast-free alternative: v_mov_b32 dpp quad_perm broadcast + fma (cost of DPP mov)
                asm volatile("v_mov_b32_dpp %3, %6 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %4, %6 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %5, %6 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_fma_f32 %0, -%6, %3, %0\n\tv_fma_f32 %1, -%6, %4, %1\n\tv_fma_f32 %2, -%6, %5, %2"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5) : "v"(l));
            else if (MODE == 6)     // 3 ds_bpermute-free: LDS broadcast read b128 (same address in every lane) + 4 fma
                asm volatile("ds_read_b128 %4, %5\n\ts_waitcnt lgkmcnt(0)\n\t"
                             "v_fma_f32 %0, -%6, %4, %0\n\tv_fma_f32 %1, -%6, %4, %1\n\tv_fma_f32 %2, -%6, %4, %2\n\tv_fma_f32 %3, -%6, %4, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(*(float __attribute__((ext_vector_type(4)))*)&p0) : "v"(0), "v"(l));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + p0[0] + p1[1] + p2[0] + s0;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
template <int MODE> void run(const char* nm, int ninstr) {
    float* o; long long* t; hipMalloc(&o, 256); hipMalloc(&t, 8);
    long long h = 0;
    for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, o, t, 1.0f); hipDeviceSynchronize(); }
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%-44s %6.2f ticks per group, %5.2f per instruction\n", nm, (double)h / (256.0 * REP), (double)h / (256.0 * REP * ninstr));
}
int main() {
    run<0>("3 readlane + 3 fma(sgpr)", 6);
    run<1>("6 fma (vgpr)", 6);
    run<2>("6 readlane", 6);
    run<3>("6 fma (sgpr operand)", 6);
    run<4>("6 readlane + 3 pk_fma", 9);
    run<5>("3 mov_dpp + 3 fma", 6);
    return 0;
}
