// Moving-average means.
//   reference: voltron/means/EWMA.py:20-37 -- conv1d(1->1 channel, k taps) over the series left-padded
//   with k copies of y[0]; output length N+1; forced to CPU fp32 there.  Here it stays on the device.
#include "common.h"
#include "../../include/volt_hip.h"

namespace volt {

// out[b,t] = sum_{j<k} w[j] * padded[b,t+j],  padded = [y0]*k ++ y,  t = 0..N.
// 256 outputs per workgroup; the k+256 inputs and the k taps sit in LDS.  fp64 accumulation
// (conv1d's summation order is unspecified; this is at least as accurate as any fp32 order).
__global__ __launch_bounds__(256) void ewma_kernel(const float* __restrict__ y, int64_t bs_y,
                                                   const float* __restrict__ w, int k, float* __restrict__ out,
                                                   int N) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sw = sm;          // k
    float* sy = sm + k;      // k + 256
    const int b = blockIdx.y, t0 = blockIdx.x * 256;
    const float* yb = y + (int64_t)b * bs_y;
    for (int j = threadIdx.x; j < k; j += 256) sw[j] = w[j];
    for (int e = threadIdx.x; e < k + 256; e += 256) {
        const int src = t0 + e - k;                       // index into y of padded[t0 + e]
        sy[e] = yb[src < 0 ? 0 : (src < N ? src : N - 1)];
    }
    __syncthreads();
    const int t = t0 + threadIdx.x;
    if (t > N) return;
    double acc = 0.0;
    for (int j = 0; j < k; ++j) acc += (double)sw[j] * (double)sy[threadIdx.x + j];
    out[(int64_t)b * (N + 1) + t] = (float)acc;
}

}  // namespace volt

extern "C" int volt_ewma_f32(const float* y, int64_t bs_y, const float* w, int k, float* out, int B, int N,
                             void* stream) {
    if (!y) return -1;
    if (!w) return -3;
    if (k < 1 || k > 16384) return -4;
    if (!out) return -5;
    if (B < 0 || B > 65535) return -6;
    if (N < 1) return -7;
    if (B == 0) return 0;
    const size_t lds = (size_t)(2 * k + 256) * sizeof(float);
    if (lds > 48 * 1024) {                        // large dynamic LDS has to be opted into (k > ~6000)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(volt::ewma_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(volt::ewma_kernel, dim3((N + 1 + 255) / 256, B), dim3(256), lds, (hipStream_t)stream, y, bs_y, w,
                       k, out, N);
    VOLT_LAUNCH_CHECK();
    return 0;
}
