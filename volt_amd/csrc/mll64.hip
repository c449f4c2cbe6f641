// fp64 marginal log likelihood + analytic gradient (7 hard part 2: "an fp64 variant with per-dtype
// tolerances").  The reference keeps the caller's dtype (voltron/kernels/VolKernel.py:28-33; gpytorch's log_prob
// computes in it), so a double-precision model trains in double precision: same step as mll.hip on the fp64 twins --
//     A = K + sigma2 I -> volt_potrf_f64 -> z = L^-1 r, alpha = L^-T z (one-launch chained solves, trsv.hip)
//     -> Y = L^-T (volt_trtri_f64) -> tr K_s^-1 = ||Y||_F^2 -> scalars.
// The O(N^2) passes are HBM streams; the O(N^3) work runs on v_mfma_f64_16x16x4_f64 (chol64.hip).
#include "common.h"
#include "tiles64.h"
#include "../../include/volt_hip.h"
#include <math.h>

size_t volt_internal_batch64_bytes(int B, int n, int has_y);   // batch64_step.hip

namespace volt {

__global__ void pad_resid64_kernel(const double* __restrict__ resid, double* __restrict__ rpad, int N, int Np) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Np) rpad[(int64_t)b * Np + i] = (i < N) ? resid[(int64_t)b * N + i] : 0.0;
}

// ||Y||_F^2 over the upper tiles, rows < N (the identity padding contributes nothing that way).  One workgroup per upper
// 128x128 tile and matrix, 16-byte loads; deterministic two-stage sum: part[b][t], t = the tile's index in row-major order
// of the upper triangle.  (Rounds 2-4 had one workgroup per 128-ROW block: 32 workgroups for one series of N = 4096 --
// 0.59 ms for a 67 MB stream, a fifth of that series' whole gradient step.)
__global__ __launch_bounds__(256) void frob64_kernel(const double* __restrict__ Y, double* __restrict__ part, int N, int Np) {
    __shared__ double red[256];
    const int n = Np / TS, b = blockIdx.y, tid = threadIdx.x;
    int rb = 0, t = blockIdx.x;                                   // tile (rb, cb), cb >= rb
    while (t >= n - rb) { t -= n - rb; ++rb; }
    const int cb = rb + t;
    const double* T = Y + (int64_t)b * Np * Np + (int64_t)rb * TS * Np + (int64_t)cb * TS;
    typedef double d2 __attribute__((ext_vector_type(2)));
    double acc = 0.0;
#pragma unroll 8
    for (int e = tid; e < TS * TS / 2; e += 256) {
        const int r = e >> 6, c = (e & 63) * 2;
        const int row = rb * TS + r, col = cb * TS + c;
        if (row < N) {
            const d2 v = *reinterpret_cast<const d2*>(T + (int64_t)r * Np + c);
            if (col >= row) acc += v[0] * v[0];
            if (col + 1 >= row) acc += v[1] * v[1];
        }
    }
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) part[(int64_t)b * (n * (n + 1) / 2) + blockIdx.x] = red[0];
}

// out[b, 0..7] = mll, dmll/dsigma2, quad, logdet, trinv, aa, sigma2-used, 0   (the layout of the fp32 step)
__global__ __launch_bounds__(256) void mll_scalars64_kernel(const double* __restrict__ A, const double* __restrict__ z,
                                                            const double* __restrict__ alpha_pad,
                                                            const double* __restrict__ frob, const double* __restrict__ sigma2,
                                                            double jitter, double* __restrict__ out,
                                                            double* __restrict__ alpha_out, int N, int Np, int want_grad) {
    __shared__ double red[256];
    const int b = blockIdx.x, tid = threadIdx.x, n = Np / TS;
    const double* Ab = A + (int64_t)b * Np * Np;
    auto block_sum = [&](double v) -> double {
        red[tid] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    double q = 0, ld = 0, aa = 0, tr = 0;
    for (int i = tid; i < N; i += 256) {
        const double zi = z[(int64_t)b * Np + i];
        q += zi * zi;
        ld += log(Ab[(int64_t)i * Np + i]);
        if (want_grad) {
            const double al = alpha_pad[(int64_t)b * Np + i];
            aa += al * al;
            alpha_out[(int64_t)b * N + i] = al;
        }
    }
    if (want_grad)
        for (int i = tid; i < n * (n + 1) / 2; i += 256) tr += frob[(int64_t)b * (n * (n + 1) / 2) + i];
    q = block_sum(q);
    ld = 2.0 * block_sum(ld);
    aa = block_sum(aa);
    tr = block_sum(tr);
    if (tid == 0) {
        const double LOG_2PI = 1.8378770664093453;
        double* o = out + (int64_t)b * 8;
        o[0] = -0.5 * (q + ld + N * LOG_2PI) / N;
        o[2] = q;
        o[3] = ld;
        o[6] = (sigma2 ? sigma2[b] : 0.0) + jitter;
        o[7] = 0.0;
        if (want_grad) {
            o[1] = 0.5 * (aa - tr) / N;
            o[4] = tr;
            o[5] = aa;
        }
    }
}

struct Mll64Ws {
    double *A, *Winv, *Y, *rpad, *z, *scratch, *apad, *frob;
    void* prog;                      // progress words of the one-launch schedule (batch64_step.hip), if the shape is its
    size_t prog_bytes, bytes;
};
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
static Mll64Ws carve64(void* base, int B, int N, int want_grad) {
    const size_t Np = (size_t)volt_padded_n(N), n = Np / TS;
    size_t off = 0;
    auto take = [&](size_t doubles) {
        double* p = base ? reinterpret_cast<double*>(reinterpret_cast<char*>(base) + off) : nullptr;
        off += al256(doubles * sizeof(double));
        return p;
    };
    Mll64Ws w;
    w.A = take((size_t)B * Np * Np);
    w.Winv = take((size_t)B * n * TS * TS);
    w.rpad = take((size_t)B * Np);
    w.z = take((size_t)B * Np);
    w.scratch = take((size_t)B * Np + 64);
    w.apad = take((size_t)B * Np);
    w.Y = want_grad ? take((size_t)B * Np * Np) : nullptr;
    w.frob = want_grad ? take((size_t)B * n * (n + 1) / 2) : nullptr;
    w.prog_bytes = volt_internal_batch64_bytes(B, (int)n, want_grad);
    w.prog = w.prog_bytes ? take(w.prog_bytes / sizeof(double)) : nullptr;
    w.bytes = off;
    return w;
}

}  // namespace volt

using namespace volt;

int volt_internal_factor_f64(double* A, double* Winv, int* info, double* Y, int B, int Np, void* stream, void* state,
                             size_t state_bytes);   // chol64.hip
int volt_internal_batch64_step(double* A, double* Winv, int* info, double* Y, int B, int Np, void* state, size_t state_bytes,
                               void* stream, const volt::KSource64* ksrc);   // batch64_step.hip

extern "C" {

size_t volt_mll_workspace_bytes_f64(int B, int N, int want_grad) {
    if (B <= 0 || N <= 0) return 0;
    return carve64(nullptr, B, N, want_grad).bytes;
}

int volt_mll_step_f64(const double* K, int64_t ldk, int64_t bsk, const double* resid, const double* sigma2, double jitter,
                      double* out, double* alpha, int* info, void* workspace, int B, int N, int want_grad, void* stream) {
    if (!K) return -1;
    if (ldk < N) return -2;
    if (!resid) return -4;
    if (!out) return -7;
    if (want_grad && !alpha) return -8;
    if (!info) return -9;
    if (!workspace || ((uintptr_t)workspace & 255)) return -10;
    if (B < 0 || B > 65535) return -11;
    if (N < 1) return -12;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int Np = volt_padded_n(N), n = Np / TS;
    Mll64Ws w = carve64(workspace, B, N, want_grad);
    int rc;
    hipLaunchKernelGGL(pad_resid64_kernel, dim3((Np + 255) / 256, B), dim3(256), 0, s, resid, w.rpad, N, Np);
    // small / medium batches: factorisation (+ inverse) as ONE launch that reads its tiles straight from K (batch64_step.hip)
    const KSource64 src{K, ldk, bsk, sigma2, jitter, N};
    rc = w.prog ? volt_internal_batch64_step(w.A, w.Winv, info, want_grad ? w.Y : nullptr, B, Np, w.prog, w.prog_bytes, stream, &src) : 0;
    if (rc != 0 && rc != 1) return rc > 0 ? rc : -1;
    // otherwise a prepared copy, then factorisation and (gradient step) the triangular inverse in one multi-stream schedule (chol64.hip)
    if (rc == 0 && (rc = volt_prepare_f64(K, ldk, bsk, sigma2, jitter, w.A, B, N, stream))) return rc > 0 ? rc : -1;
    if (rc == 0 && (rc = volt_internal_factor_f64(w.A, w.Winv, info, want_grad ? w.Y : nullptr, B, Np, stream, w.prog, w.prog_bytes))) return rc > 0 ? rc : -1;
    if ((rc = volt_trsv_lower_f64(w.A, w.Winv, w.rpad, w.z, w.scratch, B, Np, stream))) return rc > 0 ? rc : -1;
    if (want_grad) {
        if ((rc = volt_trsv_lower_t_f64(w.A, w.Winv, w.z, w.apad, w.scratch, B, Np, stream))) return rc > 0 ? rc : -1;
        hipLaunchKernelGGL(frob64_kernel, dim3(n * (n + 1) / 2, B), dim3(256), 0, s, w.Y, w.frob, N, Np);
    }
    hipLaunchKernelGGL(mll_scalars64_kernel, dim3(B), dim3(256), 0, s, w.A, w.z, w.apad, w.frob, sigma2, jitter, out, alpha,
                       N, Np, want_grad);
    VOLT_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
