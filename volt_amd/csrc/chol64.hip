// fp64 batched Cholesky (7 hard part 2 "plan an fp64 variant (fp64 MFMA)").
//   reference call sites: psd_safe_cholesky at voltron/rollout_utils.py:35 on the NOISE-FREE train block
//   (condition number 1e6 at N = 400, 1e8 at N = 4096: beyond fp32), VoltronGP.py:83, VoltMagpie.py:87; the
//   reference keeps the caller's dtype (VolKernel.py:28-33 inherits it), so an fp64 model factors in fp64.
//
// Same left-looking 128-wide block-column scheme as chol.hip, on v_mfma_f64_16x16x4_f64:
//   P1  A[i,k] -= sum_{m<k} L[i,m] L[k,m]^T     128x128 tile per workgroup, K = 128 k      (MFMA)
//   P2  L[k,k] = chol(A[k,k]),  W_k = L[k,k]^-1  one workgroup per matrix, LDS image        (VALU, latency chain)
//   P3  L[i,k] = A[i,k] W_k^T                    128x128x128 product                        (MFMA)
// A double matrix is staged as a float matrix of twice the width, so the global->register->LDS pipeline of
// common.h is reused byte for byte (a 32-float chunk row = 16 doubles = 128 B; LDS rows of 36 floats = 18
// doubles, which makes the ds_read_b64 fragment reads bank-conflict free: lane (r, k) of a 32-lane group reads
// dwords 36 r + 2 k + {0,1}, r < 16, k < 2 -- 64 distinct banks).  One 16x16x4 MFMA per (16-row, 16-column)
// pair and K step of 4; a wave owns 64x64 = 4x4 of them (128 accumulator VGPRs).
//
// Used where fp32 cannot carry the conditioning: the rollouts' train-block factor and its rho / tau
// (volt_amd/rollout_engine.py), and gp.psd_safe_cholesky on fp64 input.  Throughput matters little there
// (one factorisation per series against H x S sample steps), so P2 is a plain LDS algorithm.
#include "common.h"
#include "../../include/volt_hip.h"

namespace volt {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int SLD64 = SLD / 2;          // 18 doubles per LDS row
constexpr int BK64 = BK / 2;            // 16 doubles of K per chunk

// acc[mt*4+nt] (16x16 block at rows 16 mt, columns 16 nt of the wave's 64x64) += A rows x B rows^T over the
// K steps [KK0, KK1) of the staged chunk (4 doubles each).  Lane l supplies A[row = l & 15][k = l >> 4] and
// B[col = l & 15][k = l >> 4]; accumulator register q of lane l is element (row = (l >> 4) + 4 q, col = l & 15).
template <int KK0, int KK1>
__device__ __forceinline__ void mma_chunk64(const float* __restrict__ buf, f64x4 (&acc)[16]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lk = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const double* sA = reinterpret_cast<const double*>(buf);
    const double* sB = reinterpret_cast<const double*>(buf + TS * SLD);
#pragma unroll
    for (int kk = KK0; kk < KK1; ++kk) {
        double a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = sA[(wr * 64 + t * 16 + l15) * SLD64 + kk * 4 + lk];
            b[t] = sB[(wc * 64 + t * 16 + l15) * SLD64 + kk * 4 + lk];
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt * 4 + nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], b[nt], acc[mt * 4 + nt], 0, 0, 0);
    }
}

// acc += A[0:128, 0:16 nchunks] * B[0:128, 0:16 nchunks]^T (doubles; lda / ldb in doubles).  Same software pipeline
// as gemm_nt_128: register-staged prefetch two chunks ahead, double-buffered LDS, one barrier per chunk.
__device__ __forceinline__ void gemm64_nt_128(const double* __restrict__ A, int64_t lda, const double* __restrict__ B,
                                              int64_t ldb, int nchunks, f64x4 (&acc)[16], float* smem) {
    if (nchunks <= 0) return;
    StageRegs s0, s1;
    const StageAddr sa = stage_addr(reinterpret_cast<const float*>(A), 2 * lda, reinterpret_cast<const float*>(B), 2 * ldb);
    stage_load_buf(s0, sa, 0);
    stage_store(s0, smem);
    if (nchunks > 1) stage_load_buf(s0, sa, BK);
    if (nchunks > 2) stage_load_buf(s1, sa, 2 * BK);
    __syncthreads();
    int c = 0;
    for (; c + 1 < nchunks; c += 2) {
        float* b0 = smem;
        float* b1 = smem + STAGE_FLOATS;
        mma_chunk64<0, 2>(b0, acc);
        stage_store(s0, b1);
        if (c + 3 < nchunks) stage_load_buf(s0, sa, (c + 3) * BK);
        mma_chunk64<2, 4>(b0, acc);
        __syncthreads();
        mma_chunk64<0, 2>(b1, acc);
        if (c + 2 < nchunks) stage_store(s1, b0);
        if (c + 4 < nchunks) stage_load_buf(s1, sa, (c + 4) * BK);
        mma_chunk64<2, 4>(b1, acc);
        __syncthreads();
    }
    if (c < nchunks) {
        mma_chunk64<0, 4>(smem, acc);
        __syncthreads();
    }
}

__device__ __forceinline__ void zero_acc64(f64x4 (&acc)[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][q] = 0.0;
}

// (row, col) inside the 128x128 tile of accumulator register q of block (mt, nt) for this lane
#define VOLT_ACC64_RC(mt, nt, q)                                              \
    const int r = (wave >> 1) * 64 + (mt) * 16 + (lane >> 4) + 4 * (q);       \
    const int c = (wave & 1) * 64 + (nt) * 16 + (lane & 15);

// ----------------------------------------------------------------------------- prepare
// A = tril-tiles(K) + (sigma2 + jitter) I, identity in the padding; tile (ti, tj), tj <= ti.
__global__ __launch_bounds__(256) void prepare64_kernel(const double* __restrict__ K, int64_t ldk, int64_t bsk,
                                                        const double* __restrict__ sigma2, double jitter,
                                                        double* __restrict__ A, int N, int Np) {
    const int t = blockIdx.x;
    int ti = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    const int b = blockIdx.y;
    const double add = (sigma2 ? sigma2[b] : 0.0) + jitter;
    const double* Kb = K + (int64_t)b * bsk;
    double* Ab = A + (int64_t)b * Np * Np;
    const int c = threadIdx.x & 127;
    for (int rr = threadIdx.x >> 7; rr < TS; rr += 2) {
        const int i = ti * TS + rr, j = tj * TS + c;
        double v = (i < N && j < N) ? Kb[(int64_t)i * ldk + j] : 0.0;
        if (i == j) v = (i < N) ? v + add : 1.0;
        Ab[(int64_t)i * Np + j] = v;
    }
}

// ----------------------------------------------------------------------------- P1
// grid.x = (n-k) * B, k >= 1.  Tile t: rows of block k+t, columns of block k.
__global__ __launch_bounds__(256) void update64_kernel(double* __restrict__ A, int Np, int k, int B) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    const int n = Np / TS;
    int t, b;
    decode_tile_batch(n - k, B, t, b);
    double* Ab = A + (int64_t)b * Np * Np;
    const double* Arows = Ab + (int64_t)(k + t) * TS * Np;
    const double* Brows = Ab + (int64_t)k * TS * Np;
    double* C = Ab + (int64_t)(k + t) * TS * Np + (int64_t)k * TS;
    f64x4 acc[16];
    zero_acc64(acc);
    gemm64_nt_128(Arows, Np, Brows, Np, k * (TS / BK64), acc, smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                VOLT_ACC64_RC(mt, nt, q)
                C[(int64_t)r * Np + c] -= acc[mt * 4 + nt][q];
            }
}

// ----------------------------------------------------------------------------- P3
// grid.x = (n-k-1) * B.  L[i,k] = A[i,k] W_k^T in place (all of the tile is read before the epilogue stores).
__global__ __launch_bounds__(256) void trsm64_kernel(double* __restrict__ A, const double* __restrict__ Winv, int Np,
                                                     int k, int B) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    const int n = Np / TS;
    int t, b;
    decode_tile_batch(n - k - 1, B, t, b);
    double* P = A + (int64_t)b * Np * Np + (int64_t)(k + 1 + t) * TS * Np + (int64_t)k * TS;
    const double* W = Winv + ((int64_t)b * n + k) * TS * TS;
    f64x4 acc[16];
    zero_acc64(acc);
    gemm64_nt_128(P, Np, W, TS, TS / BK64, acc, smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                VOLT_ACC64_RC(mt, nt, q)
                P[(int64_t)r * Np + c] = acc[mt * 4 + nt][q];
            }
}

// ----------------------------------------------------------------------------- P2
// One workgroup per matrix: the 128x128 diagonal block lives in an LDS image (row stride 129 doubles: a column
// walk hits 32 distinct bank pairs).  Right-looking factorisation with ONE barrier per pivot: the rank-1 update
// of pivot j uses the still unscaled column j (a_rj a_cj / d_j), the columns are scaled in one pass at the end.
// Then W = L^-1 in place from the last column to the first (LAPACK trti2, lower):
//     W[j][j] = 1 / L[j][j],   W[r][j] = -W[j][j] sum_{m=j+1}^{r} W[r][m] L[m][j]      (r > j)
// where W[r][m], m > j, is already final and column j of L is still the original.
constexpr int DT64 = TS + 1;
constexpr int DIAG64_LDS_BYTES = (TS * DT64 + 2 * TS) * 8;

__global__ __launch_bounds__(256) void diag64_kernel(double* __restrict__ A, double* __restrict__ Winv,
                                                     int* __restrict__ info, int Np, int k) {
    extern __shared__ __attribute__((aligned(16))) double sT[];
    double* dj = sT + TS * DT64;           // pivots d_j
    double* col = dj + TS;                 // new column of W before it replaces column j
    const int n = Np / TS, b = blockIdx.x, tid = threadIdx.x;
    double* D = A + (int64_t)b * Np * Np + (int64_t)k * TS * Np + (int64_t)k * TS;
    double* W = Winv + ((int64_t)b * n + k) * TS * TS;
    for (int e = tid; e < TS * TS; e += NT) {
        const int r = e >> 7, c = e & 127;
        sT[r * DT64 + c] = (c <= r) ? D[(int64_t)r * Np + c] : 0.0;
    }
    __syncthreads();
    int bad = 0;
    const int tr = tid >> 4, tc = tid & 15;
    for (int j = 0; j < TS; ++j) {
        const double d = sT[j * DT64 + j];
        if (!(d > 0.0) && bad == 0) bad = j + 1;            // uniform: every thread reads the same pivot
        const double dinv = 1.0 / d;
        for (int r = j + 1 + tr; r < TS; r += 16) {
            const double lr = sT[r * DT64 + j] * dinv;
            for (int c = j + 1 + tc; c <= r; c += 16) sT[r * DT64 + c] -= lr * sT[c * DT64 + j];
        }
        if (tid == 0) dj[j] = d;
        __syncthreads();
    }
    // scale the columns: L[r][j] = a_rj / sqrt(d_j), L[j][j] = sqrt(d_j); L goes out (zeros above the diagonal)
    for (int e = tid; e < TS * TS; e += NT) {
        const int r = e >> 7, c = e & 127;
        double v = 0.0;
        if (c <= r) {
            const double s = sqrt(dj[c]);
            v = (c == r) ? s : sT[r * DT64 + c] / s;
            sT[r * DT64 + c] = v;
        }
        D[(int64_t)r * Np + c] = v;
    }
    __syncthreads();
    // W = L^-1 in place, last column first.  Threads r > j take one row each (128 threads; the row walk of
    // thread r and the column walk over m are both conflict free with the odd stride).
    for (int j = TS - 1; j >= 0; --j) {
        const double wjj = 1.0 / sT[j * DT64 + j];
        if (tid > j && tid < TS) {
            double a = 0.0;
            for (int m = j + 1; m <= tid; ++m) a += sT[tid * DT64 + m] * sT[m * DT64 + j];
            col[tid] = -a * wjj;
        }
        __syncthreads();
        if (tid > j && tid < TS) sT[tid * DT64 + j] = col[tid];
        if (tid == j) sT[j * DT64 + j] = wjj;
        __syncthreads();
    }
    for (int e = tid; e < TS * TS; e += NT) {
        const int r = e >> 7, c = e & 127;
        W[r * TS + c] = (c <= r) ? sT[r * DT64 + c] : 0.0;
    }
    if (tid == 0 && bad) atomicCAS(info + b, 0, k * TS + bad);
}

}  // namespace volt

using namespace volt;

extern "C" {

int volt_prepare_f64(const double* K, int64_t ldk, int64_t bsk, const double* sigma2, double jitter, double* A, int B,
                     int N, void* stream) {
    if (!K) return -1;
    if (ldk < N) return -2;
    if (!A) return -6;
    if (B < 0 || B > 65535) return -7;
    if (N < 1) return -8;
    if (B == 0) return 0;
    const int Np = volt_padded_n(N), n = Np / TS;
    hipLaunchKernelGGL(prepare64_kernel, dim3(n * (n + 1) / 2, B), dim3(256), 0, (hipStream_t)stream, K, ldk, bsk,
                       sigma2, jitter, A, N, Np);
    VOLT_LAUNCH_CHECK();
    return 0;
}

int volt_potrf_f64(double* A, double* Winv, int* info, int B, int Np, void* stream) {
    if (!A) return -1;
    if (!Winv) return -2;
    if (!info) return -3;
    if (B < 0) return -4;
    if (Np < TS || Np % TS) return -5;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int n = Np / TS;
    hipError_t e = hipMemsetAsync(info, 0, sizeof(int) * (size_t)B, s);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(diag64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            DIAG64_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    for (int k = 0; k < n; ++k) {
        if (k > 0) hipLaunchKernelGGL(update64_kernel, dim3((n - k) * B), dim3(256), 0, s, A, Np, k, B);
        hipLaunchKernelGGL(diag64_kernel, dim3(B), dim3(256), DIAG64_LDS_BYTES, s, A, Winv, info, Np, k);
        if (k + 1 < n) hipLaunchKernelGGL(trsm64_kernel, dim3((n - k - 1) * B), dim3(256), 0, s, A, Winv, Np, k, B);
    }
    VOLT_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
