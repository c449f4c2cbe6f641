// Marginal log likelihood + analytic gradient.
//   reference: loss = -mll(model(x), y); loss.backward()  -- voltron/train_utils.py:243-250,
//   :130-139, voltron/models/Volt.py:133-146; arithmetic in gpytorch ExactMarginalLogLikelihood.
// The O(N^2) passes here are HBM-bound streams over Y = L^-T (upper triangle only); the first of them
// (z = Y'r partials, Frobenius partials) is fused into the trtri epilogue in chol.hip.
#include "common.h"
#include "../../include/volt_hip.h"
#include "../../include/volt_hip_tune.h"
#include <math.h>

namespace volt {

__global__ void pad_resid_kernel(const float* __restrict__ resid, float* __restrict__ rpad, int N, int Np) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Np) rpad[(int64_t)b * Np + i] = (i < N) ? resid[(int64_t)b * N + i] : 0.f;
}

// R2: z[c] = sum_{jb <= cb} zpart[jb][c]
__global__ void sum_zpart_kernel(const float* __restrict__ zpart, float* __restrict__ z, int Np) {
    const int n = Np / TS;
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Np) return;
    const int cb = c / TS;
    float a = 0.f;
    for (int jb = 0; jb <= cb; ++jb) a += zpart[((int64_t)b * n + jb) * Np + c];
    z[(int64_t)b * Np + c] = a;
}

// R3: alpha = Y z, Y upper triangular: alpha[j] = sum_{c >= j} Y[j][c] z[c].  Pure HBM stream over the upper half of Y.
// A wave owns four rows (two at a time, one per half-wave: 512 contiguous bytes per row and 128-column block) and adds
// the blocks' dot products in ascending block order -- rowpair_dot (common.h) defines the sum, and the one-launch batched
// step's alpha items (batch_step.hip) produce the very same partials.  grid = (Np/16) * B workgroups of 4 waves.
__global__ __launch_bounds__(256) void y_times_z_kernel(const float* __restrict__ Y, const float* __restrict__ z,
                                                        float* __restrict__ alpha, int Np, int B) {
    int rb, b;
    decode_tile_batch(Np / 16, B, rb, b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
    const int j0 = rb * 16 + wave * 4;                       // this wave's 4 rows: j0 + h, j0 + 2 + h
    const float* Yb = Y + (int64_t)b * Np * Np;
    const float* zb = z + (int64_t)b * Np;
    const int n = Np / TS;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                     // rows j0, j0 + 1, j0 + 2, j0 + 3 (wave-uniform)
    const float* y0 = Yb + (int64_t)(j0 + h) * Np + 4 * l31;
    const float* y1 = Yb + (int64_t)(j0 + 2 + h) * Np + 4 * l31;
    for (int c = j0 / TS; c < n; ++c) {                      // tiles left of the diagonal tile are not stored
        const f32x4 zv = *reinterpret_cast<const f32x4*>(zb + c * TS + 4 * l31);
        const f32x4 ya = *reinterpret_cast<const f32x4*>(y0 + c * TS);
        const f32x4 yb = *reinterpret_cast<const f32x4*>(y1 + c * TS);
        float lo, hi;
        rowpair_dot(ya, zv, lo, hi);
        acc[0] += lo;
        acc[1] += hi;
        rowpair_dot(yb, zv, lo, hi);
        acc[2] += lo;
        acc[3] += hi;
    }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) alpha[(int64_t)b * Np + j0 + r] = acc[r];
    }
}

// (one-launch batched step) alpha from its partial sums per block column of Y: alpha[i] = sum_{c >= i / 128} apart[b][c][i],
// added in ascending block order -- the order y_times_z_kernel adds the same partials in, so alpha agrees bit for bit
// between the schedules.  grid (Np / 256, B): one column per thread, many workgroups -- folded into the scalars' kernel (one
// workgroup per matrix) the same sums took 22 - 37 us instead of 5 - 7 (8 x 1500, 24 x 2048): kept as a launch of its own.
__global__ __launch_bounds__(256) void alpha_sum_kernel(const float* __restrict__ apart, float* __restrict__ alpha_pad, int N,
                                                        int Np) {
    const int n = Np / TS, b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Np) return;
    float al = 0.f;
    if (i < N)
        for (int c = i / TS; c < n; ++c) al += apart[((int64_t)b * n + c) * Np + i];
    alpha_pad[(int64_t)b * Np + i] = al;
}

// R4: scalars.  out[b, 0..7] = mll, dmll/dsigma2, quad, logdet, trinv, aa, sigma2-used, 0
// One workgroup per matrix; a thread takes the columns tid + 256 u SIXTEEN AT A TIME -- the strided diagonal entries, z and
// alpha of a chunk are all requested before the first is used -- and takes ONE double log per chunk, of the product of its
// pivots (L_ii in 1e-4 .. 1e3: far inside the double range).  One column at a time this was 16 dependent round trips and 16
// software logs per thread: 13.8 us behind every step of 64 x 4096 (and 9 of 558 at 8 x 1500).
__global__ __launch_bounds__(256) void mll_scalars_kernel(const float* __restrict__ A, const float* __restrict__ z,
                                                          const float* __restrict__ alpha_pad,
                                                          const float* __restrict__ frob, const float* __restrict__ sigma2,
                                                          float jitter, float* __restrict__ out,
                                                          float* __restrict__ alpha_out, int N, int Np, int want_grad) {
    __shared__ double red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = Np / TS;
    const float* Ab = A + (int64_t)b * Np * Np;
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
    constexpr int CPT = 16;
    for (int c0 = tid; c0 < Np; c0 += CPT * 256) {
        float al[CPT], dg[CPT], zz[CPT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {                            // (clamped addresses: no branch per column)
            const int c = c0 + u * 256, cc = c < N ? c : N - 1;
            dg[u] = Ab[(int64_t)cc * Np + cc];
            zz[u] = z[(int64_t)b * Np + cc];
            al[u] = 0.f;
            if (want_grad) al[u] = alpha_pad[(int64_t)b * Np + cc];
        }
        int valid = 0;
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int c = c0 + u * 256;
            if (c < N) {
                const double zi = zz[u];
                q += zi * zi;
                valid |= 1 << u;
                if (want_grad) {
                    aa += (double)al[u] * al[u];
                    alpha_out[(int64_t)b * N + c] = al[u];
                }
            }
        }
        ld += log_pivot_product(dg, valid);                        // (NaN for a failed pivot, as sixteen single logs would give)
    }
    if (want_grad) {
        const int nt = n * (n + 1) / 2;
        for (int i = tid; i < nt; i += 256) tr += frob[(int64_t)b * nt + i];
    }
    q = block_sum(q);
    ld = 2.0 * block_sum(ld);
    aa = block_sum(aa);
    tr = block_sum(tr);
    if (tid == 0) {
        const double LOG_2PI = 1.8378770664093453;
        float* o = out + (int64_t)b * 8;
        o[0] = (float)(-0.5 * (q + ld + N * LOG_2PI) / N);
        o[2] = (float)q;
        o[3] = (float)ld;
        o[6] = (sigma2 ? sigma2[b] : 0.f) + jitter;
        o[7] = 0.f;
        if (want_grad) {
            o[1] = (float)(0.5 * (aa - tr) / N);
            o[4] = (float)tr;
            o[5] = (float)aa;
        }
    }
}

// ---- optional iterative refinement of alpha (VOLT_REFINE_ALPHA, round 4) ------------------------------------------------
// alpha <- alpha + K_s^-1 (r - K_s alpha): the residual is formed against the caller's K ITSELF with every product and the
// row sum in fp64 (K's entries and alpha are exact fp32 numbers; an fp32-accumulated residual loses the correction in its
// own rounding -- measured: up to 89x WORSE than no refinement at the noise floor, scripts/acc_diag.py), the correction is
// two triangular solves against the fp32 factor.  One pass over K (both triangles are read: the refinement needs the full
// symmetric K the reference's kernels return) -- an HBM stream, one wave per row, float4 lanes.
__device__ __forceinline__ double wave_sum_d(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}
__global__ __launch_bounds__(256) void refine_resid_kernel(const float* __restrict__ K, int64_t ldk, int64_t bsk,
                                                           const float* __restrict__ resid, const float* __restrict__ alpha,
                                                           const float* __restrict__ sigma2, float jitter,
                                                           float* __restrict__ r1pad, int N, int Np) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= Np) return;
    if (i >= N) {
        if (lane == 0) r1pad[(int64_t)b * Np + i] = 0.f;
        return;
    }
    const float* Ki = K + (int64_t)b * bsk + (int64_t)i * ldk;
    const float* ab = alpha + (int64_t)b * N;
    double acc = 0.0;
    const bool vec = ((ldk & 3) == 0) && ((bsk & 3) == 0) && (((uintptr_t)K & 15) == 0) && ((N & 3) == 0) && (((uintptr_t)alpha & 15) == 0);
    if (vec) {
        for (int j = lane * 4; j < N; j += 256) {
            const f32x4 kv = *reinterpret_cast<const f32x4*>(Ki + j);
            const f32x4 av = *reinterpret_cast<const f32x4*>(ab + j);
            acc += ((double)kv[0] * av[0] + (double)kv[1] * av[1]) + ((double)kv[2] * av[2] + (double)kv[3] * av[3]);
        }
    } else {
        for (int j = lane; j < N; j += 64) acc += (double)Ki[j] * ab[j];
    }
    acc = wave_sum_d(acc);
    if (lane == 0) {
        const double s = (double)(sigma2 ? sigma2[b] : 0.f) + (double)jitter;
        r1pad[(int64_t)b * Np + i] = (float)((double)resid[(int64_t)b * N + i] - acc - s * (double)ab[i]);
    }
}
// alpha += delta, and the scalars that depend on alpha: quad = r'alpha, a'a, mll, d mll / d sigma2
__global__ __launch_bounds__(256) void refine_finish_kernel(const float* __restrict__ delta_pad, const float* __restrict__ resid,
                                                            float* __restrict__ alpha, float* __restrict__ out, int N, int Np) {
    __shared__ double red[2][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    double q = 0, aa = 0;
    for (int i = tid; i < N; i += 256) {
        const float an = alpha[(int64_t)b * N + i] + delta_pad[(int64_t)b * Np + i];
        alpha[(int64_t)b * N + i] = an;
        q += (double)resid[(int64_t)b * N + i] * an;
        aa += (double)an * an;
    }
    red[0][tid] = q;
    red[1][tid] = aa;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double LOG_2PI = 1.8378770664093453;
        float* o = out + (int64_t)b * 8;
        o[0] = (float)(-0.5 * (red[0][0] + (double)o[3] + N * LOG_2PI) / N);
        o[1] = (float)(0.5 * (red[1][0] - (double)o[4]) / N);
        o[2] = (float)red[0][0];
        o[5] = (float)red[1][0];
        o[7] = 1.f;                                       // marks a refined alpha
    }
}

struct MllWs {
    float *A, *Winv, *Y, *rpad, *z, *scratch, *apad, *zpart, *frob, *sk_slab, *apart;
    int sk_rows;
    int* sk_count;
    void* tab;               // the balanced schedule's item tables (chol.hip), tab_bytes long
    size_t tab_bytes;
    void* small;             // state of the one-launch step for short series (chol.hip), small_bytes long
    size_t small_bytes;
    void* lng;               // state of the one-launch step for one long series (chol.hip), lng_bytes long
    size_t lng_bytes;
    float* eslab;            // ... and the slabs of its early-part slices
    void* batch;             // table + progress words of the one-launch batched step (batch_step.hip), batch_bytes long
    size_t batch_bytes;
    size_t bytes;
};

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace volt
size_t volt_internal_sched_bytes(int B, int n);   // chol.hip
int volt_internal_sched_install(void* tab, size_t tab_bytes, int B, int n, int has_y, int cap, void* stream);
size_t volt_internal_small_bytes(int B, int n);
int volt_internal_small_install(void* state, size_t bytes, int B, int n, void* stream);
int volt_internal_small_step(const float* K, int64_t ldk, int64_t bsk, const float* resid, const float* sigma2,
                             float jitter, float* A, float* Winv, float* Y, int* info, float* rpad, float* zpart,
                             float* frob, float* z, float* apad, float* apart, float* out, float* alpha, void* state,
                             int B, int N, void* stream);
size_t volt_internal_long_bytes(int B, int n);
size_t volt_internal_long_slab_floats(int B, int n);
int volt_internal_long_install(void* state, size_t bytes, int B, int n, void* stream);
int volt_internal_long_step(const float* K, int64_t ldk, int64_t bsk, const float* resid, const float* sigma2,
                            float jitter, float* A, float* Winv, float* Y, int* info, float* rpad, float* zpart,
                            float* frob, float* z, float* apad, float* apart, float* eslab, float* out, float* alpha,
                            void* state, int B, int N, void* stream);
size_t volt_internal_batch_bytes(int B, int n, int has_y);   // batch_step.hip
bool volt_internal_batch_first();
int volt_internal_batch_install(void* state, size_t bytes, int B, int n, int has_y, void* stream);
int volt_internal_batch_step(const float* K, int64_t ldk, int64_t bsk, const float* sigma2, float jitter, float* A,
                             float* Winv, float* Y, int* info, const float* rpad, float* zpart, float* frob, int B, int N,
                             float* z, float* apart, void* state, size_t state_bytes, void* stream, hipEvent_t e0,
                             hipEvent_t e1);
namespace volt {

static MllWs carve(void* base, int B, int N, int want_grad) {
    const size_t Np = (size_t)volt_padded_n(N), n = Np / TS;
    size_t off = 0;
    auto take = [&](size_t floats) {
        float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
        off += al256(floats * sizeof(float));
        return p;
    };
    MllWs w;
    w.A = take((size_t)B * Np * Np);
    w.Winv = take((size_t)B * n * TS * TS);
    w.rpad = take((size_t)B * Np);
    w.z = take((size_t)B * Np);
    w.scratch = take((size_t)B * Np);
    w.apad = take((size_t)B * Np);
    if (want_grad) {
        w.Y = take((size_t)B * Np * Np);
        w.zpart = take((size_t)B * n * Np);
        w.frob = take((size_t)B * (n * (n + 1) / 2));
    } else {
        w.Y = w.zpart = w.frob = nullptr;
    }
    // scratch of the small-batch / balanced schedules (chol.hip): slab rows of (n+1) tiles + arrival counters.  64 rows
    // below 32 matrices; the forward-only step has no trtri rows to fill the late launches with, so it gets 128 rows up to
    // 64 matrices (two slices per tile at 64)
    w.sk_rows = B < 32 ? 64 : (!want_grad && B <= 64 ? 128 : 0);
    if (w.sk_rows) {
        w.sk_slab = take((size_t)w.sk_rows * (n + 1) * TS * TS);
        w.sk_count = reinterpret_cast<int*>(take((size_t)(n + 1) * (n + 1) * B));
    } else {
        w.sk_slab = nullptr;
        w.sk_count = nullptr;
    }
    w.tab_bytes = w.sk_rows ? volt_internal_sched_bytes(B, (int)n) : 0;
    w.tab = w.tab_bytes ? take(w.tab_bytes / sizeof(float)) : nullptr;
    w.small_bytes = want_grad ? volt_internal_small_bytes(B, (int)n) : 0;
    w.small = w.small_bytes ? take(w.small_bytes / sizeof(float)) : nullptr;
    w.lng_bytes = want_grad ? volt_internal_long_bytes(B, (int)n) : 0;
    w.lng = w.lng_bytes ? take(w.lng_bytes / sizeof(float)) : nullptr;
    w.eslab = w.lng_bytes ? take(volt_internal_long_slab_floats(B, (int)n)) : nullptr;
    w.batch_bytes = volt_internal_batch_bytes(B, (int)n, want_grad);
    w.apart = (w.small_bytes || w.lng_bytes || (want_grad && w.batch_bytes)) ? take((size_t)B * n * Np) : nullptr;   // alpha's partial sums, per row of the inverse
    w.batch = w.batch_bytes ? take(w.batch_bytes / sizeof(float)) : nullptr;
    w.bytes = off;
    return w;
}

}  // namespace volt

using namespace volt;

// chol.hip: runs the factorisation group by group on the library's streams and calls `post` on each group's
// stream when that group's factor (+ inverse) is enqueued, so the O(N^2) tail of one group overlaps the other
// groups' MFMA work instead of running after the join.
typedef void (*volt_group_post_fn)(void* ctx, int b0, int Bg, hipStream_t s);
int volt_internal_factor(const float* K, int64_t ldk, int64_t bsk, const float* sigma2, float jitter, float* A,
                         float* Winv, float* Y, int* info, const float* rpad, float* zpart, float* frob, int B, int N,
                         void* stream, volt_group_post_fn post, void* post_ctx, float* sk_slab, int* sk_count, int sk_rows,
                         void* tab, size_t tab_bytes);

int volt_internal_profile(const float* K, int64_t ldk, int64_t bsk, const float* sigma2, float* A, float* Winv, float* Y,
                          int* info, const float* rpad, float* zpart, float* frob, int B, int N, int groups, void* stream,
                          float* sk_slab, int* sk_count, int sk_rows, void* tab, size_t tab_bytes, volt_group_post_fn post,
                          void* post_ctx, float* ms_sum_host, float* ms_union_host, int* launches_host,
                          float* per_launch_host);

namespace {
struct TailCtx {
    volt::MllWs w;
    const float* sigma2;
    float jitter;
    float* out;
    float* alpha;
    int N, Np, want_grad;
};

void mll_tail(void* vctx, int b0, int Bg, hipStream_t s) {
    const TailCtx& c = *static_cast<const TailCtx*>(vctx);
    const int Np = c.Np, n = Np / TS;
    const int64_t mat = (int64_t)Np * Np;
    float* z = c.w.z + (int64_t)b0 * Np;
    float* apad = c.w.apad + (int64_t)b0 * Np;
    const float* A = c.w.A + b0 * mat;
    if (c.want_grad) {
        const float* Y = c.w.Y + b0 * mat;
        hipLaunchKernelGGL(sum_zpart_kernel, dim3((Np + 255) / 256, Bg), dim3(256), 0, s,
                           c.w.zpart + (int64_t)b0 * n * Np, z, Np);
        hipLaunchKernelGGL(y_times_z_kernel, dim3((Np / 16) * Bg), dim3(256), 0, s, Y, z, apad, Np, Bg);
    } else {
        (void)volt_trsv_lower_f32(A, c.w.Winv + (int64_t)b0 * n * TS * TS, c.w.rpad + (int64_t)b0 * Np, z,
                                  c.w.scratch + (int64_t)b0 * Np, Bg, Np, (void*)s);
    }
    hipLaunchKernelGGL(mll_scalars_kernel, dim3(Bg), dim3(256), 0, s, A, z, apad,
                       c.want_grad ? c.w.frob + (int64_t)b0 * (n * (n + 1) / 2) : nullptr,
                       c.sigma2 ? c.sigma2 + b0 : nullptr, c.jitter, c.out + (int64_t)b0 * 8,
                       c.alpha ? c.alpha + (int64_t)b0 * c.N : nullptr, c.N, Np, c.want_grad);
}
// what is left behind the one-launch batched step (batch_step.hip): with the inverse, z and alpha's partial sums came out
// of the launch itself -- the scalars alone; without, the forward solve for z as above
void batch_tail(TailCtx& c, int B, hipStream_t s) {
    if (!c.want_grad) {
        mll_tail(&c, 0, B, s);
        return;
    }
    hipLaunchKernelGGL(alpha_sum_kernel, dim3((c.Np + 255) / 256, B), dim3(256), 0, s, c.w.apart, c.w.apad, c.N, c.Np);
    hipLaunchKernelGGL(mll_scalars_kernel, dim3(B), dim3(256), 0, s, c.w.A, c.w.z, c.w.apad, c.w.frob, c.sigma2, c.jitter, c.out,
                       c.alpha, c.N, c.Np, 1);
}
}  // namespace

// gpcv.hip continues from the factor and Y = L^-T this step leaves in its workspace
const float* volt_internal_mll_y(void* workspace, int B, int N) { return carve(workspace, B, N, 1).Y; }

extern "C" {

size_t volt_mll_workspace_bytes(int B, int N, int want_grad) {
    if (B <= 0 || N <= 0) return 0;
    return carve(nullptr, B, N, want_grad).bytes;
}

int volt_mll_workspace_init_f32(void* workspace, int B, int N, int want_grad, void* stream) {
    if (!workspace || ((uintptr_t)workspace & 255)) return -1;
    if (B < 1 || B > 65535) return -2;
    if (N < 1) return -3;
    MllWs w = carve(workspace, B, N, want_grad);
    const int n = volt_padded_n(N) / TS;
    if (w.small) {
        const int rc = volt_internal_small_install(w.small, w.small_bytes, B, n, stream);
        if (rc) return rc;
    }
    if (w.lng) {
        const int rc = volt_internal_long_install(w.lng, w.lng_bytes, B, n, stream);
        if (rc) return rc;
    }
    if (w.batch) {
        const int rc = volt_internal_batch_install(w.batch, w.batch_bytes, B, n, want_grad, stream);
        if (rc) return rc;
    }
    if (!w.tab) return 0;
    return volt_internal_sched_install(w.tab, w.tab_bytes, B, n, want_grad, w.sk_rows, stream);
}

int volt_mll_step_f32(const float* K, int64_t ldk, int64_t bsk, const float* resid, const float* sigma2, float jitter,
                      float* out, float* alpha, int* info, void* workspace, int B, int N, int flags,
                      void* stream) {
    const int want_grad = flags & VOLT_WANT_GRAD;
    // the caller's word that volt_mll_workspace_init_f32 ran on this workspace for this shape: only then are the regions
    // that hold tables / one-launch state handed on (and checked on the device before they are followed)
    const bool ready = (flags & VOLT_WS_INITIALISED) != 0;
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
    const int Np = volt_padded_n(N);
    MllWs w = carve(workspace, B, N, want_grad);
    int rc;
    bool done = false;
    if (ready && w.batch && volt_internal_batch_first()) {                          // batches of longer series: the whole step in one launch (batch_step.hip)
        hipLaunchKernelGGL(pad_resid_kernel, dim3((Np + 255) / 256, B), dim3(256), 0, s, resid, w.rpad, N, Np);
        rc = volt_internal_batch_step(K, ldk, bsk, sigma2, jitter, w.A, w.Winv, want_grad ? w.Y : nullptr, info, w.rpad, w.zpart,
                                      w.frob, B, N, w.z, w.apart, w.batch, w.batch_bytes, stream, nullptr, nullptr);
        if (rc == 1) {
            TailCtx ctx{w, sigma2, jitter, out, alpha, N, Np, want_grad};
            batch_tail(ctx, B, s);
            done = true;
        } else if (rc) {
            return rc > 0 ? rc : -1;
        }
    }
    if (!done && ready && want_grad && w.lng && w.apart && w.eslab) {  // one long series: one launch with sliced early parts (chol.hip)
        rc = volt_internal_long_step(K, ldk, bsk, resid, sigma2, jitter, w.A, w.Winv, w.Y, info, w.rpad, w.zpart, w.frob,
                                     w.z, w.apad, w.apart, w.eslab, out, alpha, w.lng, B, N, stream);
        if (rc == 1) done = true;
        else if (rc) return rc > 0 ? rc : -1;
    }
    if (!done && ready && want_grad && w.small && w.apart) {   // short series: the whole step in one launch (chol.hip)
        rc = volt_internal_small_step(K, ldk, bsk, resid, sigma2, jitter, w.A, w.Winv, w.Y, info, w.rpad, w.zpart, w.frob,
                                      w.z, w.apad, w.apart, out, alpha, w.small, B, N, stream);
        if (rc == 1) done = true;
        else if (rc) return rc > 0 ? rc : -1;
    }
    if (!done && ready && w.batch) {                          // batches of longer series: the whole step in one launch (batch_step.hip)
        hipLaunchKernelGGL(pad_resid_kernel, dim3((Np + 255) / 256, B), dim3(256), 0, s, resid, w.rpad, N, Np);
        rc = volt_internal_batch_step(K, ldk, bsk, sigma2, jitter, w.A, w.Winv, want_grad ? w.Y : nullptr, info, w.rpad, w.zpart,
                                      w.frob, B, N, w.z, w.apart, w.batch, w.batch_bytes, stream, nullptr, nullptr);
        if (rc == 1) {
            TailCtx ctx{w, sigma2, jitter, out, alpha, N, Np, want_grad};
            batch_tail(ctx, B, s);
            done = true;
        } else if (rc) {
            return rc > 0 ? rc : -1;
        }
    }
    if (!done) {
        hipLaunchKernelGGL(pad_resid_kernel, dim3((Np + 255) / 256, B), dim3(256), 0, s, resid, w.rpad, N, Np);
        TailCtx ctx{w, sigma2, jitter, out, alpha, N, Np, want_grad};
        if ((rc = volt_internal_factor(K, ldk, bsk, sigma2, jitter, w.A, w.Winv, want_grad ? w.Y : nullptr, info,
                                       want_grad ? w.rpad : nullptr, w.zpart, w.frob, B, N, stream, mll_tail, &ctx, w.sk_slab,
                                       w.sk_count, w.sk_rows, ready ? w.tab : nullptr, w.tab_bytes)))
            return rc > 0 ? rc : -1;
    }
    if (want_grad && (flags & VOLT_REFINE_ALPHA)) {
        // one step of iterative refinement (opt-in): fp64-accumulated residual against K, correction through the factor
        hipLaunchKernelGGL(refine_resid_kernel, dim3((Np + 3) / 4, B), dim3(256), 0, s, K, ldk, bsk, resid, alpha, sigma2, jitter,
                           w.rpad, N, Np);
        if ((rc = volt_trsv_lower_f32(w.A, w.Winv, w.rpad, w.z, w.scratch, B, Np, stream))) return rc > 0 ? rc : -1;
        if ((rc = volt_trsv_lower_t_f32(w.A, w.Winv, w.z, w.apad, w.scratch, B, Np, stream))) return rc > 0 ? rc : -1;
        hipLaunchKernelGGL(refine_finish_kernel, dim3(B), dim3(256), 0, s, w.apad, resid, alpha, out, N, Np);
    }
    VOLT_LAUNCH_CHECK();
    return 0;
}

int volt_profile_step_f32(const float* K, int64_t ldk, int64_t bsk, const float* resid, const float* sigma2,
                          float* out, float* alpha, void* workspace, int* info, int B, int N, int groups, void* stream,
                          float* ms_sum_host, float* ms_union_host, int* launches_host, float* per_launch_host) {
    if (!K) return -1;
    if (ldk < N) return -2;
    if (!resid) return -4;
    if (!out) return -6;
    if (!alpha) return -7;
    if (!workspace || ((uintptr_t)workspace & 255)) return -8;
    if (!info) return -9;
    if (B < 1 || B > 65535) return -10;
    if (N < 1) return -11;
    if (!ms_sum_host) return -14;
    if (!ms_union_host) return -15;
    if (!launches_host) return -16;
    const int Np = volt_padded_n(N);
    MllWs w = carve(workspace, B, N, 1);
    hipLaunchKernelGGL(pad_resid_kernel, dim3((Np + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, resid, w.rpad, N, Np);
    TailCtx ctx{w, sigma2, 0.f, out, alpha, N, Np, 1};
    if (groups == 0 && w.batch) {
        // the one-launch batched step (batch_step.hip) is what this shape runs: class 0 = that one launch (factorisation AND
        // inverse: the whole step's 2 N^3 / 3), class 1 empty
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return (int)hipGetLastError();
        int rc = volt_internal_batch_step(K, ldk, bsk, sigma2, 0.f, w.A, w.Winv, w.Y, info, w.rpad, w.zpart, w.frob, B, N, w.z,
                                          w.apart, w.batch, w.batch_bytes, stream, e0, e1);
        if (rc == 1) {
            batch_tail(ctx, B, (hipStream_t)stream);
            hipError_t e = hipStreamSynchronize((hipStream_t)stream);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            ms_sum_host[0] = ms_union_host[0] = ms;
            ms_sum_host[1] = ms_union_host[1] = 0.f;
            launches_host[0] = 1;
            launches_host[1] = 0;
            if (per_launch_host) per_launch_host[0] = ms;
            rc = e != hipSuccess ? (int)e : 0;
        } else if (rc == 0) {
            rc = -8;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return rc;
    }
    // groups > 0 forces that many stream groups and (like the round-2 hook) switches the small-batch schedules off
    return volt_internal_profile(K, ldk, bsk, sigma2, w.A, w.Winv, w.Y, info, w.rpad, w.zpart, w.frob, B, N, groups, stream,
                                 w.sk_slab, w.sk_count, w.sk_rows, w.tab, w.tab_bytes, mll_tail, &ctx, ms_sum_host,
                                 ms_union_host, launches_host, per_launch_host);
}

}  // extern "C"
