// Covariance assembly: CumTrapz prefix integral and the min(i,j) gather.
//   reference: voltron/kernels/VolKernel.py:4-10 (CumTrapz), :18-42 (VolatilityKernel.forward)
#include "common.h"
#include "../../include/volt_hip.h"

namespace volt {

// One workgroup per series.  The running sum is inherently ordered if it is to be bit-identical to
// the reference's CPU cumsum (fp64 accumulator, each prefix rounded to T), so one lane walks the
// series out of LDS; the products w*y are formed by all lanes with coalesced loads.  O(N) work on
// 16 KB per series -- nowhere near any roofline, it only has to be exact.
template <typename T>
__global__ __launch_bounds__(256) void cumtrapz_kernel(const T* __restrict__ vol, int64_t bs_vol,
                                                       const T* __restrict__ x, int64_t bs_x,
                                                       T* __restrict__ V, int N, int square) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* prod = reinterpret_cast<T*>(smem_raw);
    const int b = blockIdx.x;
    const T* v = vol + (int64_t)b * bs_vol;
    const T* xb = x + (int64_t)b * bs_x;
    const T dx = xb[1] - xb[0];
    const T half = dx * T(0.5);
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        T y = v[i];
        if (square) y = y * y;                       // vol_path * vol_path  (VolKernel.py:28)
        const T w = (i == 0 || i == N - 1) ? half : dx;
        // explicit single rounding of the product (no contraction with the double add below)
        if constexpr (sizeof(T) == 4) prod[i] = __fmul_rn(w, y);
        else prod[i] = __dmul_rn(w, y);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double acc = 0.0;
        for (int i = 0; i < N; ++i) {
            acc += (double)prod[i];
            prod[i] = (T)acc;
        }
    }
    __syncthreads();
    T* out = V + (int64_t)b * N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) out[i] = prod[i];
}

// K[b,i,j] = V[b,min(i,j)].  Pure HBM-write stream: 4*N^2 bytes out, 4*N in (L2-resident).
// Workgroup = 16 rows x 256 columns; a thread owns one 16-byte column quad and walks 4 rows, so
// every store instruction of a wave writes 64 x 16 B = 1 KiB contiguous.  Rows per thread measured at 64 x 4096^2
// (scripts/fill_tune.sh, best of 5): 2 -> 6.57, 4 -> 6.54, 8 -> 6.30, 16 -> 5.75, 32 -> 4.95 TB/s: short workgroups win.  Non-temporal stores:
// the matrix is far larger than L2 and is not re-read by this kernel.
#ifndef VOLT_FILL_ROWS
#define VOLT_FILL_ROWS 4
#endif
constexpr int FILL_RPT = VOLT_FILL_ROWS;          // rows per thread; a workgroup covers 4 * FILL_RPT rows x 256 columns
template <typename T, int VEC>
__global__ __launch_bounds__(256) void fill_kernel(const T* __restrict__ V, T* __restrict__ K, int N,
                                                   int64_t ldk, int64_t bsk) {
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    const int b = blockIdx.z;
    const T* v = V + (int64_t)b * N;
    T* k = K + (int64_t)b * bsk;
    const int j0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * VEC;
    const int i0 = blockIdx.y * (4 * FILL_RPT) + (threadIdx.x >> 6);
    if (j0 >= N) return;
    T vj[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) vj[c] = (j0 + c < N) ? v[j0 + c] : T(0);
#pragma unroll
    for (int r = 0; r < FILL_RPT; ++r) {
        const int i = i0 + 4 * r;
        if (i >= N) break;
        const T vi = v[i];
        vec_t o;
#pragma unroll
        for (int c = 0; c < VEC; ++c) o[c] = (j0 + c <= i) ? vj[c] : vi;
        T* dst = k + (int64_t)i * ldk + j0;
        if (j0 + VEC <= N) {
            __builtin_nontemporal_store(o, reinterpret_cast<vec_t*>(dst));
        } else {
            for (int c = 0; c < VEC && j0 + c < N; ++c) dst[c] = o[c];
        }
    }
}

template <typename T>
static int launch_cumtrapz(const T* vol, int64_t bs_vol, const T* x, int64_t bs_x, T* V, int B, int N,
                           int square, void* stream) {
    if (!vol) return -1;
    if (!x) return -3;
    if (!V) return -5;
    if (B < 0) return -6;
    if (N < 2 || (size_t)N * sizeof(T) > 160 * 1024) return -7;   // x[1]-x[0] needs N >= 2
    if (B == 0) return 0;
    if ((size_t)N * sizeof(T) > 48 * 1024) {      // large dynamic LDS has to be opted into
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cumtrapz_kernel<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)N * sizeof(T)));
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(cumtrapz_kernel<T>, dim3(B), dim3(256), (size_t)N * sizeof(T), (hipStream_t)stream, vol,
                       bs_vol, x, bs_x, V, N, square);
    VOLT_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int launch_fill(const T* V, T* K, int B, int N, int64_t ldk, int64_t bsk, void* stream) {
    if (!V) return -1;
    if (!K) return -2;
    if (B < 0 || B > 65535) return -3;          // the batch rides in gridDim.z
    if (N < 1) return -4;
    if (ldk < N) return -5;
    if (B == 0) return 0;
    constexpr int VEC = 16 / sizeof(T);
    const bool aligned = (ldk % VEC == 0) && (bsk % VEC == 0) && ((uintptr_t)K % 16 == 0);
    if (aligned) {
        dim3 grid((N + 64 * VEC - 1) / (64 * VEC), (N + 4 * FILL_RPT - 1) / (4 * FILL_RPT), B);
        hipLaunchKernelGGL((fill_kernel<T, VEC>), grid, dim3(256), 0, (hipStream_t)stream, V, K, N, ldk, bsk);
    } else {
        dim3 grid((N + 63) / 64, (N + 4 * FILL_RPT - 1) / (4 * FILL_RPT), B);
        hipLaunchKernelGGL((fill_kernel<T, 1>), grid, dim3(256), 0, (hipStream_t)stream, V, K, N, ldk, bsk);
    }
    VOLT_LAUNCH_CHECK();
    return 0;
}

}  // namespace volt

extern "C" {

int volt_abi_version(void) { return VOLT_ABI_VERSION; }
#ifndef VOLT_SOURCE_HASH
#define VOLT_SOURCE_HASH "unknown"
#endif
const char* volt_source_hash(void) { return VOLT_SOURCE_HASH; }
int volt_padded_n(int n) { return ((n + volt::TS - 1) / volt::TS) * volt::TS; }

int volt_cumtrapz_f32(const float* vol, int64_t bs_vol, const float* x, int64_t bs_x, float* V, int B, int N,
                      int square, void* stream) {
    return volt::launch_cumtrapz<float>(vol, bs_vol, x, bs_x, V, B, N, square, stream);
}
int volt_cumtrapz_f64(const double* vol, int64_t bs_vol, const double* x, int64_t bs_x, double* V, int B, int N,
                      int square, void* stream) {
    return volt::launch_cumtrapz<double>(vol, bs_vol, x, bs_x, V, B, N, square, stream);
}
int volt_fill_f32(const float* V, float* K, int B, int N, int64_t ldk, int64_t bsk, void* stream) {
    return volt::launch_fill<float>(V, K, B, N, ldk, bsk, stream);
}
int volt_fill_f64(const double* V, double* K, int B, int N, int64_t ldk, int64_t bsk, void* stream) {
    return volt::launch_fill<double>(V, K, B, N, ldk, bsk, stream);
}

}  // extern "C"
