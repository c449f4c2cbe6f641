// What ONE wave can issue per clock: 2048 independent VALU instructions (16 accumulators round-robin), once as straight-line code
// and once as a 32-instruction loop body run 64 times -- separating the VALU's issue rate from what instruction fetch delivers to
// a single wave.  (s_memtime counts at 2.4 GHz.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rate scripts/ubench/issue_rate.hip && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define I0(i) "v_fmac_f64_e32 v[" #i "*2+2:" #i "*2+3], v[40:41], v[42:43]\n\t"
#define I1(i) "v_fma_f64 v[" #i "*2+2:" #i "*2+3], -v[40:41], v[42:43], v[" #i "*2+2:" #i "*2+3]\n\t"
#define I2(i) "v_fmac_f32_e32 v[" #i "+2], v40, v42\n\t"
#define I3(i) "v_fma_f32 v[" #i "+2], -v40, v42, v[" #i "+2]\n\t"
#define I4(i) "v_fmac_f64_dpp v[" #i "*2+2:" #i "*2+3], v[40:41], v[42:43] row_newbcast:" #i " row_mask:0xf bank_mask:0xf\n\t"
#define I5(i) "v_readlane_b32 s[" #i "+40], v40, " #i "\n\t"
#define I6(i) "v_fmac_f64_e32 v[" #i "*2+2:" #i "*2+3], s[60:61], v[42:43]\n\t"
#define I7(i) "v_fma_f64 v[" #i "*2+2:" #i "*2+3], -v[40:41], s[60:61], v[" #i "*2+2:" #i "*2+3]\n\t"
#define I8(i) "v_mov_b32_dpp v[" #i "+2], v40 row_newbcast:" #i " row_mask:0xf bank_mask:0xf\n\t"
#define CLOB "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v40","v41","v42","v43","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s60","s61"
#define B32(I) R16(I) R16(I)
#define B256(I) B32(I) B32(I) B32(I) B32(I) B32(I) B32(I) B32(I) B32(I)
#define B2048(I) B256(I) B256(I) B256(I) B256(I) B256(I) B256(I) B256(I) B256(I)
#define INIT "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0x3ff00000\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0x3f500000\n\ts_mov_b32 s60, 0\n\ts_mov_b32 s61, 0x3ff00000\n\t"

template <int M, bool LOOP>
__global__ void k(long long* t) {
    asm volatile(INIT ::: CLOB);
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (LOOP) {
#pragma unroll 1
        for (int it = 0; it < 64; ++it) {
            if (M == 0) asm volatile(B32(I0) ::: CLOB); if (M == 1) asm volatile(B32(I1) ::: CLOB); if (M == 2) asm volatile(B32(I2) ::: CLOB);
            if (M == 3) asm volatile(B32(I3) ::: CLOB); if (M == 4) asm volatile(B32(I4) ::: CLOB); if (M == 5) asm volatile(B32(I5) ::: CLOB);
            if (M == 6) asm volatile(B32(I6) ::: CLOB); if (M == 7) asm volatile(B32(I7) ::: CLOB); if (M == 8) asm volatile(B32(I8) ::: CLOB);
        }
    } else {
        if (M == 0) asm volatile(B2048(I0) ::: CLOB); if (M == 1) asm volatile(B2048(I1) ::: CLOB); if (M == 2) asm volatile(B2048(I2) ::: CLOB);
        if (M == 3) asm volatile(B2048(I3) ::: CLOB); if (M == 4) asm volatile(B2048(I4) ::: CLOB); if (M == 5) asm volatile(B2048(I5) ::: CLOB);
        if (M == 6) asm volatile(B2048(I6) ::: CLOB); if (M == 7) asm volatile(B2048(I7) ::: CLOB); if (M == 8) asm volatile(B2048(I8) ::: CLOB);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
template <int M> void run(const char* nm, int bytes) {
    long long* t; (void)hipMalloc(&t, 8);
    long long h0 = 0, h1 = 0;
    for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL((k<M, false>), dim3(1), dim3(64), 0, 0, t); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&h0, t, 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL((k<M, true>), dim3(1), dim3(64), 0, 0, t); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&h1, t, 8, hipMemcpyDeviceToHost);
    printf("%-44s %d bytes   straight-line %5.2f clocks each   32-instruction loop %5.2f clocks each\n", nm, bytes, h0 / 2048.0, h1 / 2048.0);
}
int main() {
    run<0>("v_fmac_f64_e32 (VGPRs)", 4);
    run<1>("v_fma_f64 with a neg modifier (VOP3)", 8);
    run<6>("v_fmac_f64_e32 with an SGPR pair", 4);
    run<7>("v_fma_f64 neg, SGPR pair (tiles64.h today)", 8);
    run<4>("v_fmac_f64_dpp row_newbcast", 8);
    run<2>("v_fmac_f32_e32", 4);
    run<3>("v_fma_f32 with a neg modifier (VOP3)", 8);
    run<5>("v_readlane_b32", 8);
    run<8>("v_mov_b32_dpp row_newbcast", 8);
    return 0;
}
