// GPCV volatility extraction: one ELBO + gradient step of the variational GP.
//   reference: LearnGPCV, voltron/train_utils.py:15-67 -- loss = -VariationalELBO(...)(model(train_x), yy);
//   model voltron/models/single_task_variational_gp.py:69-122, likelihood
//   voltron/likelihoods/volatility_likelihood.py:42-50; the ELBO arithmetic itself is gpytorch's.
// With inducing points == inputs and the unwhitened strategy the latent distribution is q(u) = N(m, Lq Lq')
// itself, so the step is
//     ell  = sum_i GH_75[ log N(y_i; 0, max(exp f, 1e-3)) ],  f ~ N(m_i, sum_j Lq_ij^2)        (O(N^2) stream)
//     KL   = 1/2 ( |L^-1 Lq|_F^2 + |L^-1 (m - mu)|^2 - N + logdet K - logdet S ),  K = L L'      (dense, MFMA)
// and its gradients  dKL/dLq = tril(K^-1 Lq) - diag(1/Lq_ii),  dKL/dm = K^-1 (m - mu) = -dKL/dmu,
//     dKL/dK = 1/2 (K^-1 - G G' - beta beta'),  G = K^-1 Lq.
// The factorisation, Y = L^-T, beta, tr K^-1 and logdet K come from the exact-GP step (mll.hip); what is new
// here is a batched structured "NT" GEMM on the same 128x128 MFMA core for T' = Lq' L^-T and G = Y T'.
#include "common.h"
#include "../../include/volt_hip.h"
#include <math.h>

namespace volt {

// dst (Np x Np, zero padded) = transpose of the kept triangle of src (N x N, leading dim lds):
// keep = 1: src lower (col <= row), keep = 2: src upper (col >= row), 0: everything.
__global__ __launch_bounds__(256) void transpose_tri_kernel(const float* __restrict__ src, int64_t lds, int64_t bss,
                                                            float* __restrict__ dst, int N, int Np, int keep) {
    __shared__ float t[32][33];
    const int b = blockIdx.z;
    const float* s = src + (int64_t)b * bss;
    float* d = dst + (int64_t)b * Np * Np;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;            // source tile origin
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        float v = 0.f;
        if (r < N && c < N && (keep == 0 || (keep == 1 && c <= r) || (keep == 2 && c >= r))) v = s[(int64_t)r * lds + c];
        t[ty + 8 * k][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = c0 + ty + 8 * k, c = r0 + tx;                   // destination (row, col) = source (col, row)
        d[(int64_t)r * Np + c] = t[tx][ty + 8 * k];
    }
}

// C[b] tile (tm, tn) = alpha * A[b][tm, :] B[b][tn, :]^T + beta * C, K restricted by the operands' triangles:
// s = 0 dense, 1 lower (k-block <= row-block), 2 upper (k-block >= row-block).  sc: 0 every tile, 1 only
// tn <= tm, 2 only tn >= tm (other tiles are not touched).  frob (nullable) [B, mt*nt] receives each
// tile's sum of squares (0 for skipped tiles).  All dimensions are multiples of 128.
struct GemmArgs {
    const float *A, *B;
    float* C;
    int64_t lda, bsa, ldb, bsb, ldc, bsc;
    int mt, nt, kt, sa, sb, sc;
    float alpha, beta;
    float* frob;
};

__global__ __launch_bounds__(NT, 2) void gemm_nt_struct_kernel(GemmArgs g, int nbatch) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    int tile, b;
    decode_tile_batch(g.mt * g.nt, nbatch, tile, b);
    const int tm = tile / g.nt, tn = tile % g.nt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
    const bool skip = (g.sc == 1 && tn > tm) || (g.sc == 2 && tn < tm);
    float ss = 0.f;
    if (!skip) {
        int k0 = 0, k1 = g.kt;
        if (g.sa == 1) k1 = min(k1, tm + 1);
        if (g.sa == 2) k0 = max(k0, tm);
        if (g.sb == 1) k1 = min(k1, tn + 1);
        if (g.sb == 2) k0 = max(k0, tn);
        f32x16 acc[4];
        zero_acc(acc);
        const float* Ar = g.A + (int64_t)b * g.bsa + (int64_t)tm * TS * g.lda + (int64_t)k0 * TS;
        const float* Br = g.B + (int64_t)b * g.bsb + (int64_t)tn * TS * g.ldb + (int64_t)k0 * TS;
        gemm_nt_128<0>(Ar, g.lda, Br, g.ldb, (k1 - k0) * (TS / BK), acc, smem);
        float* C = g.C + (int64_t)b * g.bsc + (int64_t)tm * TS * g.ldc + (int64_t)tn * TS;
#pragma unroll
        for (int im = 0; im < 2; ++im)
#pragma unroll
            for (int in = 0; in < 2; ++in)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int r = wr * 64 + im * 32 + accrow(q, lane);
                    const int c = wc * 64 + in * 32 + (lane & 31);
                    float v = g.alpha * acc[im * 2 + in][q];
                    if (g.beta != 0.f) v += g.beta * C[(int64_t)r * g.ldc + c];
                    C[(int64_t)r * g.ldc + c] = v;
                    ss += v * v;
                }
    }
    if (g.frob) {
        float* red = smem;
        __syncthreads();
        const float w = wave_sum_f(ss);
        if (lane == 0) red[wave] = w;
        __syncthreads();
        if (threadIdx.x == 0) g.frob[(int64_t)b * g.mt * g.nt + tile] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

static int launch_gemm(const GemmArgs& g, int B, hipStream_t s) {
    hipLaunchKernelGGL(gemm_nt_struct_kernel, dim3(g.mt * g.nt * B), dim3(NT), 0, s, g, B);
    VOLT_LAUNCH_CHECK();
    return 0;
}

// Expected log likelihood row by row.  One wave per row i: var_i = sum_{j<=i} Lq_ij^2 (coalesced row read), then
// lane k takes quadrature nodes k, k+64: f = m_i + sqrt(2 var) x_k, s = max(exp f, min_scale),
//   logp = -y^2 / (2 s^2) - log s - log sqrt(2 pi),  dlogp/df = (y^2/s^2 - 1) [exp f > min_scale].
// rowstat[b][i] = { E[logp], dE/dm_i, dE/dvar_i, log Lq_ii^2 }.   w are the hermgauss weights / sqrt(pi).
__global__ __launch_bounds__(256) void gh_ell_kernel(const float* __restrict__ m, const float* __restrict__ Lq,
                                                     const float* __restrict__ y, const float* __restrict__ ghx,
                                                     const float* __restrict__ ghw, int Q, float min_var,
                                                     float min_scale, float* __restrict__ rowstat, int N) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    const float* row = Lq + ((int64_t)b * N + i) * N;
    float s2 = 0.f;
    for (int j = lane; j <= i; j += 64) {
        const float v = row[j];
        s2 += v * v;
    }
    float var = wave_sum_f(s2);
    const bool floored = var < min_var;
    if (floored) var = min_var;
    const float mi = m[(int64_t)b * N + i], yi = y[(int64_t)b * N + i];
    const float sd2 = sqrtf(2.f * var);
    float e = 0.f, gm = 0.f, gv = 0.f;
    for (int k = lane; k < Q; k += 64) {
        const float xk = ghx[k], wk = ghw[k];
        const float f = mi + sd2 * xk;
        const float ef = expf(f);
        const bool live = ef > min_scale;
        const float sc = live ? ef : min_scale;
        const float r = yi / sc;
        const float logp = -0.5f * r * r - (live ? f : logf(min_scale)) - 0.91893853320467274f;
        const float g = live ? (r * r - 1.f) : 0.f;
        e += wk * logp;
        gm += wk * g;
        gv += wk * g * xk;
    }
    e = wave_sum_f(e);
    gm = wave_sum_f(gm);
    gv = wave_sum_f(gv);
    if (lane == 0) {
        const float dii = row[i];
        float* o = rowstat + ((int64_t)b * N + i) * 4;
        o[0] = e;
        o[1] = gm;
        o[2] = floored ? 0.f : gv / sd2;                 // df/dvar = x_k / sqrt(2 var)
        o[3] = logf(dii * dii);
    }
}

// Gradient of F = we ell - wk KL:
//   dF/dLq[i,j] = we 2 gv_i Lq_ij - wk (G_ij - [i == j] / Lq_ii)   (j <= i, zero above),
//   dF/dm = we gm - wk beta,  dF/dmu = wk beta.
__global__ __launch_bounds__(256) void gpcv_grad_kernel(const float* __restrict__ Lq, const float* __restrict__ G,
                                                        const float* __restrict__ rowstat,
                                                        const float* __restrict__ beta, float* __restrict__ gLq,
                                                        float* __restrict__ gm, float* __restrict__ gmu, int N,
                                                        int Np, float we, float wk) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const int64_t e = ((int64_t)b * N + i) * N + j;
    const float gv = rowstat[((int64_t)b * N + i) * 4 + 2];
    float v = 0.f;
    if (j <= i) {
        const float l = Lq[e];
        float kl = G[((int64_t)b * Np + i) * Np + j];
        if (j == i) kl -= 1.f / l;
        v = we * 2.f * gv * l - wk * kl;
    }
    gLq[e] = v;
    if (j == 0) {
        const float be = beta[(int64_t)b * N + i];
        gm[(int64_t)b * N + i] = we * rowstat[((int64_t)b * N + i) * 4 + 1] - wk * be;
        gmu[(int64_t)b * N + i] = wk * be;
    }
}

// dF/dK = -wk/2 (K^-1 - G G' - beta beta')  from P = K^-1 - G G' (padded) and beta.
__global__ __launch_bounds__(256) void gpcv_dk_kernel(const float* __restrict__ P, const float* __restrict__ beta,
                                                      float* __restrict__ gK, int N, int Np, float wk) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const float bi = beta[(int64_t)b * N + i], bj = beta[(int64_t)b * N + j];
    gK[((int64_t)b * N + i) * N + j] = -0.5f * wk * (P[((int64_t)b * Np + i) * Np + j] - bi * bj);
}

// out[b, 0..11] = ell, KL, quad, logdet K, logdet S, tr(K^-1 S), tr K^-1, |G|_F^2, |beta|^2, we ell - wk KL, jitter, 0
__global__ __launch_bounds__(256) void gpcv_scalars_kernel(const float* __restrict__ rowstat,
                                                           const float* __restrict__ mllout,
                                                           const float* __restrict__ frobT,
                                                           const float* __restrict__ frobG, float jitter,
                                                           float* __restrict__ out, int N, int ntiles, float we,
                                                           float wk) {
    __shared__ double red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
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
    double ell = 0, lds = 0, tr = 0, gg = 0;
    for (int i = tid; i < N; i += 256) {
        ell += rowstat[((int64_t)b * N + i) * 4];
        lds += rowstat[((int64_t)b * N + i) * 4 + 3];
    }
    for (int i = tid; i < ntiles; i += 256) {
        tr += frobT[(int64_t)b * ntiles + i];
        gg += frobG[(int64_t)b * ntiles + i];
    }
    ell = block_sum(ell);
    lds = block_sum(lds);
    tr = block_sum(tr);
    gg = block_sum(gg);
    if (tid == 0) {
        const float* mo = mllout + (int64_t)b * 8;
        const double quad = mo[2], ldk = mo[3];
        const double kl = 0.5 * (tr + quad - N + ldk - lds);
        float* o = out + (int64_t)b * 12;
        o[0] = (float)ell;
        o[1] = (float)kl;
        o[2] = (float)quad;
        o[3] = (float)ldk;
        o[4] = (float)lds;
        o[5] = (float)tr;
        o[6] = mo[4];
        o[7] = (float)gg;
        o[8] = mo[5];
        o[9] = (float)(we * ell - wk * kl);
        o[10] = jitter;
        o[11] = 0.f;
    }
}

static inline size_t al256g(size_t x) { return (x + 255) & ~(size_t)255; }

struct GpcvWs {
    float *LqT, *W, *Tt, *G, *P, *mllout, *beta, *rowstat, *frobT, *frobG;
    void* mll;
    size_t bytes;
};

static GpcvWs carve_gpcv(void* base, int B, int N, int want_dk) {
    const size_t Np = (size_t)volt_padded_n(N), n = Np / TS;
    size_t off = al256g(volt_mll_workspace_bytes(B, N, 1));
    auto take = [&](size_t floats) {
        float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
        off += al256g(floats * sizeof(float));
        return p;
    };
    GpcvWs w;
    w.mll = base;
    w.LqT = take((size_t)B * Np * Np);
    w.W = take((size_t)B * Np * Np);
    w.Tt = take((size_t)B * Np * Np);
    w.G = take((size_t)B * Np * Np);
    w.P = want_dk ? take((size_t)B * Np * Np) : nullptr;
    w.mllout = take((size_t)B * 8);
    w.beta = take((size_t)B * N);
    w.rowstat = take((size_t)B * N * 4);
    w.frobT = take((size_t)B * n * n);
    w.frobG = take((size_t)B * n * n);
    w.bytes = off;
    return w;
}

}  // namespace volt

using namespace volt;

const float* volt_internal_mll_y(void* workspace, int B, int N);      // mll.hip

extern "C" {

int volt_gemm_nt_f32(const float* A, int64_t lda, int64_t bsa, int uplo_a, const float* Bm, int64_t ldb, int64_t bsb,
                     int uplo_b, float* Cm, int64_t ldc, int64_t bsc, int uplo_c, float alpha, float beta, int batch,
                     int M, int N, int K, void* stream) {
    if (!A) return -1;
    if (uplo_a < 0 || uplo_a > 2) return -4;
    if (!Bm) return -5;
    if (uplo_b < 0 || uplo_b > 2) return -8;
    if (!Cm) return -9;
    if (uplo_c < 0 || uplo_c > 2) return -12;
    if (batch < 0) return -15;
    if (M <= 0 || M % TS) return -16;
    if (N <= 0 || N % TS) return -17;
    if (K <= 0 || K % TS) return -18;
    if (lda < K || lda % 4 || ((uintptr_t)A & 15)) return -2;
    if (ldb < K || ldb % 4 || ((uintptr_t)Bm & 15)) return -6;
    if (ldc < N) return -10;
    if (bsa % 4) return -3;
    if (bsb % 4) return -7;
    if (batch == 0) return 0;
    GemmArgs g{A, Bm, Cm, lda, bsa, ldb, bsb, ldc, bsc, M / TS, N / TS, K / TS, uplo_a, uplo_b, uplo_c, alpha, beta, nullptr};
    return launch_gemm(g, batch, (hipStream_t)stream);
}

size_t volt_gpcv_workspace_bytes(int B, int N, int want_dk) {
    if (B <= 0 || N <= 0) return 0;
    return carve_gpcv(nullptr, B, N, want_dk).bytes;
}

int volt_gpcv_step_f32(const float* K, int64_t ldk, int64_t bsk, float jitter, const float* resid, const float* m,
                       const float* Lq, const float* y, const float* gh_x, const float* gh_w, int Q, float min_var,
                       float min_scale, float w_ell, float w_kl, float* out, float* grad_m, float* grad_mu,
                       float* grad_Lq, float* grad_K, int* info, void* workspace, int B, int N, int ws_flags, void* stream) {
    if (!K) return -1;
    if (ldk < N) return -2;
    if (!resid) return -5;
    if (!m) return -6;
    if (!Lq) return -7;
    if (!y) return -8;
    if (!gh_x) return -9;
    if (!gh_w) return -10;
    if (Q < 1 || Q > 1024) return -11;
    if (!out) return -16;
    if (!grad_m) return -17;
    if (!grad_mu) return -18;
    if (!grad_Lq) return -19;
    if (!info) return -21;
    if (!workspace || ((uintptr_t)workspace & 255)) return -22;
    if (B < 0 || B > 65535) return -23;
    if (N < 1) return -24;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int Np = volt_padded_n(N), n = Np / TS;
    const int64_t mat = (int64_t)Np * Np;
    const int want_dk = grad_K != nullptr;
    GpcvWs w = carve_gpcv(workspace, B, N, want_dk);

    // the O(N^2) likelihood rows do not depend on the factorisation: enqueue them first
    hipLaunchKernelGGL(gh_ell_kernel, dim3((N + 3) / 4, B), dim3(256), 0, s, m, Lq, y, gh_x, gh_w, Q, min_var, min_scale,
                       w.rowstat, N);
    hipLaunchKernelGGL(transpose_tri_kernel, dim3(Np / 32, Np / 32, B), dim3(256), 0, s, Lq, (int64_t)N,
                       (int64_t)N * N, w.LqT, N, Np, 1);
    // K + jitter I = L L',  Y = L^-T,  beta = K^-1 resid,  quad, logdet K, tr K^-1, |beta|^2  (exact-GP step)
    int rc = volt_mll_step_f32(K, ldk, bsk, resid, nullptr, jitter, w.mllout, w.beta, info, w.mll, B, N,
                               VOLT_WANT_GRAD | (ws_flags & VOLT_WS_INITIALISED), stream);
    if (rc) return rc;
    const float* Y = volt_internal_mll_y(workspace, B, N);
    hipLaunchKernelGGL(transpose_tri_kernel, dim3(Np / 32, Np / 32, B), dim3(256), 0, s, Y, (int64_t)Np, mat, w.W, Np, Np,
                       2);
    // T' = Lq' L^-T : rows of Lq' (upper) against rows of W = L^-1 (lower); upper triangle of tiles only
    GemmArgs g1{w.LqT, w.W, w.Tt, Np, mat, Np, mat, Np, mat, n, n, n, 2, 1, 2, 1.f, 0.f, w.frobT};
    if ((rc = launch_gemm(g1, B, s))) return rc;
    // G = K^-1 Lq = Y T : rows of Y (upper) against rows of T' (upper)
    GemmArgs g2{Y, w.Tt, w.G, Np, mat, Np, mat, Np, mat, n, n, n, 2, 2, 0, 1.f, 0.f, w.frobG};
    if ((rc = launch_gemm(g2, B, s))) return rc;
    hipLaunchKernelGGL(gpcv_grad_kernel, dim3((N + 255) / 256, N, B), dim3(256), 0, s, Lq, w.G, w.rowstat, w.beta,
                       grad_Lq, grad_m, grad_mu, N, Np, w_ell, w_kl);
    if (want_dk) {
        GemmArgs g3{Y, Y, w.P, Np, mat, Np, mat, Np, mat, n, n, n, 2, 2, 0, 1.f, 0.f, nullptr};       // K^-1 = Y Y'
        if ((rc = launch_gemm(g3, B, s))) return rc;
        GemmArgs g4{w.G, w.G, w.P, Np, mat, Np, mat, Np, mat, n, n, n, 0, 0, 0, -1.f, 1.f, nullptr};   // - G G'
        if ((rc = launch_gemm(g4, B, s))) return rc;
        hipLaunchKernelGGL(gpcv_dk_kernel, dim3((N + 255) / 256, N, B), dim3(256), 0, s, w.P, w.beta, grad_K, N, Np, w_kl);
    }
    hipLaunchKernelGGL(gpcv_scalars_kernel, dim3(B), dim3(256), 0, s, w.rowstat, w.mllout, w.frobT, w.frobG, jitter, out,
                       N, n * n, w_ell, w_kl);
    VOLT_LAUNCH_CHECK();
    return 0;
}

int volt_mll_grad_k_f32(void* mll_workspace, const float* alpha, float* scratch, float* grad_K, int B, int N,
                        void* stream) {
    if (!mll_workspace || ((uintptr_t)mll_workspace & 255)) return -1;
    if (!alpha) return -2;
    if (!scratch || ((uintptr_t)scratch & 15)) return -3;
    if (!grad_K) return -4;
    if (B < 0 || B > 65535) return -5;
    if (N < 1) return -6;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int Np = volt_padded_n(N), n = Np / TS;
    const int64_t mat = (int64_t)Np * Np;
    const float* Y = volt_internal_mll_y(mll_workspace, B, N);
    GemmArgs g{Y, Y, scratch, Np, mat, Np, mat, Np, mat, n, n, n, 2, 2, 0, 1.f, 0.f, nullptr};       // K_s^-1 = Y Y'
    int rc = launch_gemm(g, B, s);
    if (rc) return rc;
    hipLaunchKernelGGL(gpcv_dk_kernel, dim3((N + 255) / 256, N, B), dim3(256), 0, s, scratch, alpha, grad_K, N, Np,
                       1.f / (float)N);
    VOLT_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
