// Issue-rate microbenchmark of v_mfma_f64_16x16x4_f64 (and v_mfma_f32_32x32x2_f32 beside it): no memory traffic,
// W waves per SIMD, 8 independent accumulators per wave.  Prints TFLOP/s for the whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k64(double* out, int iters) {
    f64x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f64x4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k32(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* o64; float* o32;
    hipMalloc(&o64, 8 << 20); hipMalloc(&o32, 8 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;                    // 256-thread blocks: 4 waves, one per SIMD
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL(k64, dim3(blocks), dim3(256), 0, 0, o64, iters); hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("f64 16x16x4  %d wave(s)/SIMD: %.1f TFLOP/s\n", wps, (double)blocks * 4 * iters * 8 * 2048 / ms / 1e9);
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, o32, iters); hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("f32 32x32x2  %d wave(s)/SIMD: %.1f TFLOP/s\n", wps, (double)blocks * 4 * iters * 4 * 4096 / ms / 1e9);
        }
    }
    return 0;
}
