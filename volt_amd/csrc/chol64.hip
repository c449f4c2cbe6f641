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
// (volt_amd/rollout_engine.py), and gp.psd_safe_cholesky on fp64 input.
#include "common.h"
#include "tiles64.h"
#include "host.h"
#include "../../include/volt_hip.h"
#include "../../include/volt_hip_tune.h"
#include <mutex>
#include <stdlib.h>

namespace volt {


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
// A[row0 + t, k] -= sum_{m = kb0}^{kb1-1} L[row0 + t, m] L[k, m]^T for t = 0 .. ntiles-1.   grid = (ntiles * B, 1, S).
// S > 1 cuts the K range into S slices, one workgroup each, which subtract their partial products with hardware fp64
// atomics (global_atomic_add_f64): few matrices leave most CUs idle in the late block columns, where a launch has
// 8 (n - k) tiles of K = 128 k each (8 x 4096, k = 28: 32 tiles on 256 CUs).  The order in which the slices land is
// not fixed, so with S > 1 the last bits of the factor can differ from run to run (the fp32 path's slab scheme is
// bitwise repeatable; in fp64 the spread is ~1e-16 relative and the tests hold 1e-9 .. 1e-11).
__global__ __launch_bounds__(256, 2) void update64_kernel(double* __restrict__ A, int Np, int k, int row0, int ntiles, int kb0,
                                                       int kb1, int B, int S, int atomic) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    const int nx = ntiles * B, len = kb1 - kb0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // grid = nx * S workgroups (or fewer, striding over the (tile, slice) pairs: persistent bulk grids of <= 248 workgroups,
    // one per CU, that leave CUs free for the chain's kernels were measured -- 8 x 4096 potrf 6.58 ms either way -- and are not used)
    for (int v = blockIdx.x; v < nx * S; v += gridDim.x) {
        int t, b;
        decode_tile_batch(v % nx, ntiles, B, t, b);
        const int sl = v / nx;
        const int c0 = kb0 + sl * len / S, c1 = kb0 + (sl + 1) * len / S;
        if (c1 <= c0) continue;
        double* Ab = A + (int64_t)b * Np * Np;
        const double* Arows = Ab + (int64_t)(row0 + t) * TS * Np + (int64_t)c0 * TS;
        const double* Brows = Ab + (int64_t)k * TS * Np + (int64_t)c0 * TS;
        double* C = Ab + (int64_t)(row0 + t) * TS * Np + (int64_t)k * TS;
        f64x4 acc[16];
        zero_acc64(acc);
        gemm64_nt_128(Arows, Np, Brows, Np, (c1 - c0) * (TS / BK64), acc, smem);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    VOLT_ACC64_RC(mt, nt, q)
                    if (!atomic) C[(int64_t)r * Np + c] -= acc[mt * 4 + nt][q];
                    else unsafeAtomicAdd(&C[(int64_t)r * Np + c], -acc[mt * 4 + nt][q]);
                }
    }
}

// ----------------------------------------------------------------------------- P3
// grid.x = (n-k-1) * B.  L[i,k] = A[i,k] W_k^T in place (all of the tile is read before the epilogue stores).
__global__ __launch_bounds__(256, 2) void trsm64_kernel(double* __restrict__ A, const double* __restrict__ Winv, int Np,
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

// ----------------------------------------------------------------------------- P2 (tiles64.h: diag64_body)
template <bool STAMP>
__global__ __launch_bounds__(256) void diag64_kernel(double* __restrict__ A, double* __restrict__ Winv,
                                                     int* __restrict__ info, int Np, int k, long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) double sT[];
    diag64_body<STAMP>(A, Winv, info, Np, k, blockIdx.x, sT, stamps, false);
}

// ----------------------------------------------------------------------------- trtri (fp64)
// Y = L^-T (upper, row-major) by block rows of X = L^-1, as in chol.hip:  X[i,j] = -W_i sum_{m=j}^{i-1} L[i,m] X[m,j].
// Two launches per block row i, the intermediate through the UNUSED lower-triangle slots of the Y buffer:
//   phase 1  S(i,j)[c][p] = sum_k Y[j-rows c][k] L[i-rows p][k],  k over columns [128 j, 128 i)   -> slot (i, j) of Y
//   phase 2  Y[j-rows c][i-cols r] = -sum_p S(i,j)[c][p] W_i[r][p];  and Y[i,i] = W_i^T
// Both are "NT" products of K-contiguous rows on the 128x128 fp64 core.  (The fp32 path keeps S in the accumulators
// and fuses the phases, chol.hip; here the tile makes one round trip through HBM -- N^2/2 doubles per matrix.)
// K range of tile j: blocks [max(j, klo), khi) -- the whole sum is (0, i); the look-ahead schedule cuts it into the early
// part (0, i-1), which needs rows <= i-2 of the inverse only, and the last block (i-1, i).  `atomic`: add to the slot
// (zeroed once) instead of storing -- K slices, and the two parts of the look-ahead schedule.
__global__ __launch_bounds__(256, 2) void trtri64_p1_kernel(const double* __restrict__ A, double* __restrict__ Y, int Np, int i,
                                                         int ntiles, int B, int klo, int khi, int atomic) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    int j, b;
    decode_tile_batch(ntiles, B, j, b);
    // K slices (gridDim.z > 1: the slices add their partial products with fp64 atomics -- few matrices would otherwise
    // leave a launch as long as its longest tile on one CU, as in update64_kernel)
    const int k0 = j > klo ? j : klo;
    const int kb = khi - k0;
    if (kb <= 0) return;
    int nsl = gridDim.z < (kb + 1) / 2 ? gridDim.z : (kb + 1) / 2;
    if (nsl < 1) nsl = 1;
    const int sl = blockIdx.z;
    if (sl >= nsl) return;
    const int c0 = k0 + sl * kb / nsl, c1 = k0 + (sl + 1) * kb / nsl;
    const double* Ab = A + (int64_t)b * Np * Np;
    double* Yb = Y + (int64_t)b * Np * Np;
    f64x4 acc[16];
    zero_acc64(acc);
    gemm64_nt_128(Yb + (int64_t)j * TS * Np + (int64_t)c0 * TS, Np, Ab + (int64_t)i * TS * Np + (int64_t)c0 * TS, Np,
                  (c1 - c0) * (TS / BK64), acc, smem);
    double* S = Yb + (int64_t)i * TS * Np + (int64_t)j * TS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                VOLT_ACC64_RC(mt, nt, q)
                if (!atomic) S[(int64_t)r * Np + c] = acc[mt * 4 + nt][q];
                else unsafeAtomicAdd(&S[(int64_t)r * Np + c], acc[mt * 4 + nt][q]);
            }
}

// zero the strictly-lower tiles of Y (the slots the sliced phase 1 accumulates into); grid (n (n-1) / 2, B)
__global__ __launch_bounds__(256) void trtri64_zero_kernel(double* __restrict__ Y, int Np) {
    const int t = blockIdx.x;
    int ti = (int)((sqrtf(8.f * (float)t + 1.f) + 1.f) * 0.5f);
    while (ti * (ti - 1) / 2 > t) --ti;
    while ((ti + 1) * ti / 2 <= t) ++ti;
    const int tj = t - ti * (ti - 1) / 2;                          // tj < ti
    double* T = Y + (int64_t)blockIdx.y * Np * Np + (int64_t)ti * TS * Np + (int64_t)tj * TS;
    const f64x2 z = {0.0, 0.0};
    for (int e = threadIdx.x; e < TS * TS / 2; e += NT) {
        const int r = e >> 6, c = (e & 63) * 2;
        *reinterpret_cast<f64x2*>(T + (int64_t)r * Np + c) = z;
    }
}

// grid: (i + 1) * B; tile j == i transposes W_i into Y[i,i]
__global__ __launch_bounds__(256, 2) void trtri64_p2_kernel(const double* __restrict__ Winv, double* __restrict__ Y, int Np,
                                                         int i, int B) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    const int n = Np / TS;
    int j, b;
    decode_tile_batch(i + 1, B, j, b);
    double* Yb = Y + (int64_t)b * Np * Np;
    const double* W = Winv + ((int64_t)b * n + i) * TS * TS;
    if (j == i) {
        trtri64_diag_body(W, Yb + (int64_t)i * TS * Np + (int64_t)i * TS, Np, smem);
        return;
    }
    const double* S = Yb + (int64_t)i * TS * Np + (int64_t)j * TS;
    f64x4 acc[16];
    zero_acc64(acc);
    gemm64_nt_128(S, Np, W, TS, TS / BK64, acc, smem);
    double* Out = Yb + (int64_t)j * TS * Np + (int64_t)i * TS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                VOLT_ACC64_RC(mt, nt, q)
                Out[(int64_t)r * Np + c] = -acc[mt * 4 + nt][q];
            }
}

}  // namespace volt

using namespace volt;

// chol.hip: the library's stream pool (one auxiliary stream, fork event, two more events, the enqueue mutex)
struct VoltAux {
    hipStream_t aux, aux2, aux3, aux4;
    hipEvent_t fork, ev[12];
    std::mutex* mu;
};
bool volt_internal_aux(VoltAux* out);

// Experiment knobs: compiled-in defaults unless the process was started with VOLT_TUNE=1 (like tunables() in chol.hip)
static int tune_int(const char* name, int dflt) {
    static const bool on = [] { const char* t = getenv("VOLT_TUNE"); return t && atoi(t) != 0; }();
    if (!on) return dflt;
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

#define VOLT_TRY64(call)                            \
    do {                                            \
        hipError_t e__ = (call);                    \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

// The GEMM kernels fit two workgroups per CU (72 KB of LDS, 246 registers; one staging register set, gemm64_nt_128) and
// run at 72 TF/s that way when a launch has many rounds of workgroups (64 x 4096, update launch k = 16 alone on the chip;
// 64 - 66 TF/s with one workgroup per CU; 56 - 58 with the two staging sets that allowed only one).  A launch of up to 512
// workgroups is SPREAD OUT instead -- 16 KB of LDS padding, one workgroup per CU, two rounds -- : the dispatcher otherwise
// pairs them up, two share one MFMA pipe and the launch lasts as long as the slower pair (8 x 4096 potrf / inverse / MLL
// step, ms: no spreading 6.38 / 6.05 / 11.7, up to 256 5.93 / 6.05 / 11.5, **up to 512 5.64 / 4.77 / 9.57**; flat beyond).
// (slot-count gates are for the full chip: scaled with the device's CU count, host.h)
static int per_chip(int v) { return (int)((int64_t)v * tunables().cus / 256); }
static unsigned spread64(int workgroups) {
    static const int lim = per_chip(tune_int("VOLT_F64_SPREAD", 512));
    return workgroups <= lim ? 16 * 1024 : 0;
}
static int trtri64_slices(int i, int B) {                       // row i: i B tiles of 1 .. i K blocks
    static const int target = per_chip(tune_int("VOLT_F64_SPLIT_TARGET", 512));
    int S = target / (i * B);
    if (S > (i + 1) / 2) S = (i + 1) / 2;
    if (S > 16) S = 16;
    return S < 1 ? 1 : S;
}
// One-row look-ahead of the inverse (trtri64_early / trtri64_finish on two streams).  Measured at N = 4096, ms without -> with:
// volt_trtri_f64 alone B = 1 2.10 -> 1.88, 2: 2.88 -> 2.60, 4: 4.05 -> 3.87, 8: 5.18 -> 5.14, 16: 9.0 -> 9.1; inside the MLL step
// B <= 4 -1 %, B = 8 10.3 -> 12.0, 16: 17.6 -> 20.3 (from 8 matrices on the inverse is bound by the 128x128 fp64 core's
// throughput, not by its chain, and one more stream only gets in the factorisation's way): up to 4 matrices.
static bool trtri64_lookahead(int B) {
    static const int env = tune_int("VOLT_F64_TRTRI_LOOKAHEAD", -1);
    return env >= 0 ? env != 0 : B <= 4;
}
static void trtri64_begin(double* Y, int B, int Np, hipStream_t s, bool lookahead = false) {
    const int n = Np / TS;
    bool any = lookahead;
    for (int i = 1; i < n; ++i) any = any || trtri64_slices(i, B) > 1;
    if (any) hipLaunchKernelGGL(trtri64_zero_kernel, dim3(n * (n - 1) / 2, B), dim3(256), 0, s, Y, Np);
}
static void trtri64_row(const double* A, const double* Winv, double* Y, int B, int Np, int i, hipStream_t s) {
    if (i > 0) {
        const int S = trtri64_slices(i, B);
        hipLaunchKernelGGL(trtri64_p1_kernel, dim3(i * B, 1, S), dim3(256), spread64((i * B) * (S)), s, A, Y, Np, i, i, B, 0, i, S > 1);
    }
    hipLaunchKernelGGL(trtri64_p2_kernel, dim3((i + 1) * B), dim3(256), spread64((i + 1) * B), s, Winv, Y, Np, i, B);
}
// The look-ahead form of a row (slots zeroed by trtri64_begin(.., true)): the early part of phase 1 -- every K block but
// the last, which needs rows <= i-2 of the inverse only -- goes out a row ahead on its own stream ...
static void trtri64_early(const double* A, double* Y, int B, int Np, int i, hipStream_t s) {
    if (i < 2) return;
    const int S = trtri64_slices(i - 1, B);
    hipLaunchKernelGGL(trtri64_p1_kernel, dim3((i - 1) * B, 1, S), dim3(256), spread64((i - 1) * B * S), s, A, Y, Np, i, i - 1, B, 0,
                       i - 1, 1);
}
// ... and the chain of the inverse is two one-block launches per row: the last block of phase 1, then phase 2
static void trtri64_finish(const double* A, const double* Winv, double* Y, int B, int Np, int i, hipStream_t s) {
    if (i > 0) hipLaunchKernelGGL(trtri64_p1_kernel, dim3(i * B, 1, 1), dim3(256), spread64((i * B) * (1)), s, A, Y, Np, i, i, B, i - 1, i, 1);
    hipLaunchKernelGGL(trtri64_p2_kernel, dim3((i + 1) * B), dim3(256), spread64((i + 1) * B), s, Winv, Y, Np, i, B);
}

// Factorisation (+ optional triangular inverse Y = L^-T, row k-1 riding on a THIRD stream beside block column k: at
// small batches the latency chain of the factorisation leaves most CUs idle, and the inverse fills them).
int volt_internal_factor_f64(double* A, double* Winv, int* info, double* Y, int B, int Np, void* stream, void* state = nullptr,
                             size_t state_bytes = 0);
// batch64_step.hip: the one-launch schedule of small batches (state: the progress words, caller scratch)
bool volt_internal_batch64_applies(int B, int n, int has_y);
size_t volt_internal_batch64_bytes(int B, int n, int has_y);
size_t volt_internal_batch64_trtri_bytes(int B, int n);
int volt_internal_batch64_trtri(const double* A, const double* Winv, double* Y, int B, int Np, void* state, size_t state_bytes,
                                void* stream);
int volt_internal_batch64_step(double* A, double* Winv, int* info, double* Y, int B, int Np, void* state, size_t state_bytes,
                               void* stream, const volt::KSource64* ksrc = nullptr);

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

int volt_tune_diag_f64(double* A, double* Winv, int* info, int B, int Np, int k, long long* stamps, void* stream) {
    if (!A) return -1;
    if (!Winv) return -2;
    if (!info) return -3;
    if (B < 1) return -4;
    if (Np < TS || Np % TS) return -5;
    if (k < 0 || k >= Np / TS) return -6;
    if (!stamps) return -7;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(diag64_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, DIAG64_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(diag64_kernel<true>, dim3(B), dim3(256), DIAG64_LDS_BYTES, (hipStream_t)stream, A, Winv, info, Np, k, stamps);
    VOLT_LAUNCH_CHECK();
    return 0;
}

int volt_trtri_f64(const double* A, const double* Winv, double* Y, int B, int Np, void* stream) {
    if (!A) return -1;
    if (!Winv) return -2;
    if (!Y) return -3;
    if (B < 0) return -4;
    if (Np < TS || Np % TS) return -5;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int n = Np / TS;
    const bool look = trtri64_lookahead(B);
    VoltAux ax;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool two = look && n >= 3 && volt_internal_aux(&ax) && hipStreamIsCapturing(s, &cap) == hipSuccess &&
                     cap == hipStreamCaptureStatusNone;
    if (!two) {
        trtri64_begin(Y, B, Np, s);
        for (int i = 0; i < n; ++i) trtri64_row(A, Winv, Y, B, Np, i, s);
        VOLT_LAUNCH_CHECK();
        return 0;
    }
    // Row i's early part (stream aux) runs beside row i-1's last block and phase 2 (the caller's stream)
    std::lock_guard<std::mutex> lock(*ax.mu);
    hipEvent_t ev_p2[2] = {ax.ev[7], ax.ev[8]}, ev_p1a = ax.ev[9];
    trtri64_begin(Y, B, Np, s, true);
    VOLT_TRY64(hipEventRecord(ax.fork, s));
    VOLT_TRY64(hipStreamWaitEvent(ax.aux, ax.fork, 0));
    for (int i = 0; i < n; ++i) {
        if (i >= 2) {
            VOLT_TRY64(hipStreamWaitEvent(ax.aux, ev_p2[i & 1], 0));        // row i-2 of the inverse is out
            trtri64_early(A, Y, B, Np, i, ax.aux);
            VOLT_TRY64(hipEventRecord(ev_p1a, ax.aux));
            VOLT_TRY64(hipStreamWaitEvent(s, ev_p1a, 0));
        }
        trtri64_finish(A, Winv, Y, B, Np, i, s);
        if (i + 2 < n) VOLT_TRY64(hipEventRecord(ev_p2[i & 1], s));
    }
    VOLT_LAUNCH_CHECK();
    return 0;
}

size_t volt_trtri_workspace_bytes_f64(int B, int Np) {
    if (B <= 0 || Np < TS || Np % TS) return 0;
    return volt_internal_batch64_trtri_bytes(B, Np / TS);
}

int volt_trtri_ws_f64(const double* A, const double* Winv, double* Y, int B, int Np, void* ws, size_t ws_bytes, void* stream) {
    if (!A) return -1;
    if (!Winv) return -2;
    if (!Y) return -3;
    if (B < 0) return -4;
    if (Np < TS || Np % TS) return -5;
    if (ws && ((uintptr_t)ws & 255)) return -6;
    if (B == 0) return 0;
    if (ws) {                                                // the whole inverse in one launch (batch64_step.hip)
        const int rc = volt_internal_batch64_trtri(A, Winv, Y, B, Np, ws, ws_bytes, stream);
        if (rc == 1) return 0;
        if (rc != 0) return rc;
    }
    return volt_trtri_f64(A, Winv, Y, B, Np, stream);
}

int volt_potrf_f64(double* A, double* Winv, int* info, int B, int Np, void* stream) {
    return volt_internal_factor_f64(A, Winv, info, nullptr, B, Np, stream);
}

size_t volt_potrf_workspace_bytes_f64(int B, int Np) {
    if (B <= 0 || Np < TS || Np % TS) return 0;
    return volt_internal_batch64_bytes(B, Np / TS, 0);
}

int volt_potrf_ws_f64(double* A, double* Winv, int* info, int B, int Np, void* ws, size_t ws_bytes, void* stream) {
    if (ws && ((uintptr_t)ws & 255)) return -6;
    return volt_internal_factor_f64(A, Winv, info, nullptr, B, Np, stream, ws, ws_bytes);
}

int volt_potrf_k_f64(const double* K, int64_t ldk, int64_t bsk, const double* sigma2, double jitter, double* A, double* Winv,
                     int* info, int B, int N, void* ws, size_t ws_bytes, void* stream) {
    if (!K) return -1;
    if (ldk < N) return -2;
    if (!A) return -6;
    if (!Winv) return -7;
    if (!info) return -8;
    if (B < 0) return -9;
    if (N < 1) return -10;
    if (ws && ((uintptr_t)ws & 255)) return -11;
    if (B == 0) return 0;
    const int Np = volt_padded_n(N);
    if (ws) {                                                         // one launch, the tiles straight from K
        const KSource64 src{K, ldk, bsk, sigma2, jitter, N};
        const int rc = volt_internal_batch64_step(A, Winv, info, nullptr, B, Np, ws, ws_bytes, stream, &src);
        if (rc != 0) return rc == 1 ? 0 : rc;
    }
    const int rc = volt_prepare_f64(K, ldk, bsk, sigma2, jitter, A, B, N, stream);
    if (rc) return rc;
    return volt_internal_factor_f64(A, Winv, info, nullptr, B, Np, stream, ws, ws_bytes);
}

}  // extern "C"

int volt_internal_factor_f64(double* A, double* Winv, int* info, double* Y, int B, int Np, void* stream, void* state,
                             size_t state_bytes) {
    if (!A) return -1;
    if (!Winv) return -2;
    if (!info) return -3;
    if (B < 0) return -4;
    if (Np < TS || Np % TS) return -5;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int n = Np / TS;
    if (state) {                                                      // a small batch: the whole schedule as one launch
        const int rc = volt_internal_batch64_step(A, Winv, info, Y, B, Np, state, state_bytes, stream);
        if (rc != 0) return rc == 1 ? 0 : rc;
    }
    hipError_t e = hipMemsetAsync(info, 0, sizeof(int) * (size_t)B, s);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(diag64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            DIAG64_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    // Left-looking with a ONE-COLUMN LOOK-AHEAD on a second stream.  What column k needs from column k-1 is one K block
    // (m = k-1); everything older can be applied a column early.  The caller's stream walks only the latency chain
    //     C(k):  last block into the DIAGONAL tile (k,k)  ->  diagonal block (128 dependent pivots, W_k)  ->  panel solve
    // while the auxiliary stream does the work that is merely wide:
    //     A(k):  last block into the other tiles (i,k), i > k;  then blocks m < k into ALL tiles of column k+1
    // (the bulk of the flops, K-sliced with fp64 atomics so that a launch has ~512 workgroups whatever the batch).
    // Events: a = column done (C -> A), c = panel tiles ready for their solve (A -> C), b[2] = column k+1's old blocks
    // applied (A -> next C; two of them because A(k) is enqueued before C(k) has waited for A(k-1)'s).
    // look-ahead depth: 0 = one stream, 1 = one column, 2 = two columns.  Measured (N = 4096, potrf / MLL step, ms, depth 1 ->
    // depth 2): B = 8 6.55 -> 5.87 / 10.6 -> 10.2, B = 16 11.6 -> 9.7 / 19.2 -> 17.6, 32 x 2048 3.81 -> 3.28 / 6.82 -> 6.54; B <= 4
    // potrf +-2 % and the step 1 - 5 % SLOWER (the third stream competes with the rows of the inverse): depth 2 from B = 6
    static const int look_env = tune_int("VOLT_F64_LOOKAHEAD", -1);
    const int look = look_env >= 0 ? look_env : (B >= 6 ? 2 : 1);
    static const int target = per_chip(tune_int("VOLT_F64_SPLIT_TARGET", 512));
    auto slices = [&](int tiles, int kblocks) {
        int S = tiles > 0 ? target / tiles : 1;
        if (S > kblocks / 2) S = kblocks / 2;                          // a slice is at least two K blocks long
        if (S > 16) S = 16;
        return S < 1 ? 1 : S;
    };
    VoltAux ax;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool two = look && n >= 3 && volt_internal_aux(&ax) && hipStreamIsCapturing(s, &cap) == hipSuccess;
    if (!two) {
        for (int k = 0; k < n; ++k) {
            if (k > 0) {
                const int S = slices((n - k) * B, k);
                hipLaunchKernelGGL(update64_kernel, dim3((n - k) * B * S), dim3(256), spread64((n - k) * B * S), s, A, Np, k, k, n - k, 0, k, B, S, S > 1);
            }
            hipLaunchKernelGGL(diag64_kernel<false>, dim3(B), dim3(256), DIAG64_LDS_BYTES, s, A, Winv, info, Np, k, nullptr);
            if (k + 1 < n) hipLaunchKernelGGL(trsm64_kernel, dim3((n - k - 1) * B), dim3(256), spread64((n - k - 1) * B), s, A, Winv, Np, k, B);
        }
        if (Y) {
            trtri64_begin(Y, B, Np, s);
            for (int i = 0; i < n; ++i) trtri64_row(A, Winv, Y, B, Np, i, s);
        }
        VOLT_LAUNCH_CHECK();
        return 0;
    }
    std::lock_guard<std::mutex> lock(*ax.mu);
    hipEvent_t ev_a = ax.ev[0], ev_c = ax.ev[1], ev_d = ax.ev[4];
    hipEvent_t ev_b[2] = {ax.ev[2], ax.ev[3]};
    VOLT_TRY64(hipEventRecord(ax.fork, s));
    VOLT_TRY64(hipStreamWaitEvent(ax.aux, ax.fork, 0));
    // Rows of the inverse beside the factorisation, themselves with a one-row look-ahead: call k (column k-1 is complete)
    // puts the last block + phase 2 of row k-1 on aux2 and the early part of row k -- it needs rows <= k-2 only -- on aux4
    hipEvent_t ev_p2[2] = {ax.ev[7], ax.ev[8]}, ev_p1a = ax.ev[9];
    // (not inside a graph capture: the fourth branch it adds crashed hipStreamEndCapture here; captured steps keep the plain rows)
    const bool tri_look = trtri64_lookahead(B) && cap == hipStreamCaptureStatusNone;
    if (Y) {
        VOLT_TRY64(hipStreamWaitEvent(ax.aux2, ax.fork, 0));
        if (tri_look) VOLT_TRY64(hipStreamWaitEvent(ax.aux4, ax.fork, 0));   // (a forked stream must get work that is joined: captures)
        trtri64_begin(Y, B, Np, ax.aux2, tri_look);
    }
    auto tri_step = [&](int k) -> int {
        VOLT_TRY64(hipStreamWaitEvent(ax.aux2, ev_a, 0));
        if (!tri_look) {
            trtri64_row(A, Winv, Y, B, Np, k - 1, ax.aux2);
            return 0;
        }
        if (k - 1 >= 2) VOLT_TRY64(hipStreamWaitEvent(ax.aux2, ev_p1a, 0));
        trtri64_finish(A, Winv, Y, B, Np, k - 1, ax.aux2);
        VOLT_TRY64(hipEventRecord(ev_p2[(k - 1) & 1], ax.aux2));
        if (k >= 2 && k < n) {
            VOLT_TRY64(hipStreamWaitEvent(ax.aux4, ev_p2[k & 1], 0));              // row k-2 is out
            trtri64_early(A, Y, B, Np, k, ax.aux4);
            VOLT_TRY64(hipEventRecord(ev_p1a, ax.aux4));
        }
        return 0;
    };
    if (look >= 2) {
        // TWO-column look-ahead: the wide part of a column's update gets two chain steps to finish in instead of one.
        // Column j receives   Z(j-2): blocks m <= j-3   (third stream, enqueued when column j-3 is done)
        //                     Y(j-1): block  m  = j-2   (auxiliary stream, when column j-2 is done)
        //                     X(j):   block  m  = j-1 into the tiles below the diagonal (auxiliary stream), and the
        //                             same block into the diagonal tile on the chain itself.
        // X, Y and Z of different steps may touch one tile at the same time: all three subtract with fp64 atomics.
        hipEvent_t ev_y = ax.ev[2], ev_z[2] = {ax.ev[3], ax.ev[5]};
        if (n >= 4) VOLT_TRY64(hipStreamWaitEvent(ax.aux3, ax.fork, 0));     // Z(1) exists from four block columns on
        for (int k = 0; k < n; ++k) {
            // the chain's waits first (they refer to what earlier iterations recorded)
            if (k >= 2) VOLT_TRY64(hipStreamWaitEvent(s, ev_y, 0));                 // Y(k-1): block k-2 is in column k
            if (k >= 3) VOLT_TRY64(hipStreamWaitEvent(s, ev_z[k & 1], 0));          // Z(k-2): blocks <= k-3 are in column k
            if (k >= 1 && k + 1 < n) {
                VOLT_TRY64(hipStreamWaitEvent(ax.aux, ev_a, 0));                    // column k-1 complete
                hipLaunchKernelGGL(update64_kernel, dim3((n - k - 1) * B), dim3(256), spread64((n - k - 1) * B), ax.aux, A, Np, k, k + 1, n - k - 1,
                                   k - 1, k, B, 1, 1);                               // X(k)
                VOLT_TRY64(hipEventRecord(ev_c, ax.aux));
                hipLaunchKernelGGL(update64_kernel, dim3((n - k - 1) * B), dim3(256), spread64((n - k - 1) * B), ax.aux, A, Np, k + 1, k + 1,
                                   n - k - 1, k - 1, k, B, 1, 1);                    // Y(k): block k-1 into column k+1
                VOLT_TRY64(hipEventRecord(ev_y, ax.aux));
            }
            if (k >= 1 && k + 2 < n) {
                VOLT_TRY64(hipStreamWaitEvent(ax.aux3, ev_a, 0));
                const int S = slices((n - k - 2) * B, k);
                hipLaunchKernelGGL(update64_kernel, dim3((n - k - 2) * B * S), dim3(256), spread64((n - k - 2) * B * S), ax.aux3, A, Np,
                                   k + 2, k + 2, n - k - 2, 0, k, B, S, 1);          // Z(k): blocks <= k-1 into column k+2
                VOLT_TRY64(hipEventRecord(ev_z[k & 1], ax.aux3));
            }
            if (Y && k >= 1) { const int rc_ = tri_step(k); if (rc_) return rc_; }
            if (k >= 1)
                hipLaunchKernelGGL(update64_kernel, dim3(B), dim3(256), spread64(B), s, A, Np, k, k, 1, k - 1, k, B, 1, 1);   // block k-1 into (k,k)
            hipLaunchKernelGGL(diag64_kernel<false>, dim3(B), dim3(256), DIAG64_LDS_BYTES, s, A, Winv, info, Np, k, nullptr);
            if (k + 1 < n) {
                if (k >= 1) VOLT_TRY64(hipStreamWaitEvent(s, ev_c, 0));
                hipLaunchKernelGGL(trsm64_kernel, dim3((n - k - 1) * B), dim3(256), spread64((n - k - 1) * B), s, A, Winv, Np, k, B);
                VOLT_TRY64(hipEventRecord(ev_a, s));
            }
        }
    } else
    for (int k = 0; k < n; ++k) {
        // ---- A(k) on the auxiliary stream (needs column k-1 complete: event a)
        if (k >= 1) {
            VOLT_TRY64(hipStreamWaitEvent(ax.aux, ev_a, 0));
            if (k + 1 < n) {
                hipLaunchKernelGGL(update64_kernel, dim3((n - k - 1) * B), dim3(256), spread64((n - k - 1) * B), ax.aux, A, Np, k, k + 1,
                                   n - k - 1, k - 1, k, B, 1, 0);             // block k-1 into the tiles (i,k), i > k
                VOLT_TRY64(hipEventRecord(ev_c, ax.aux));
                const int S = slices((n - k - 1) * B, k);
                hipLaunchKernelGGL(update64_kernel, dim3((n - k - 1) * B * S), dim3(256), spread64((n - k - 1) * B * S), ax.aux, A, Np, k + 1,
                                   k + 1, n - k - 1, 0, k, B, S, S > 1);     // blocks m < k into every tile of column k+1
                VOLT_TRY64(hipEventRecord(ev_b[k & 1], ax.aux));
            }
        }
        // ---- row k-1 of the triangular inverse on the third stream (needs column k-1 and W_{k-1}: event a)
        if (Y && k >= 1) { const int rc_ = tri_step(k); if (rc_) return rc_; }
        // ---- C(k) on the caller's stream
        if (k >= 1) {
            if (k >= 2) VOLT_TRY64(hipStreamWaitEvent(s, ev_b[(k - 1) & 1], 0));   // column k's old blocks are in
            hipLaunchKernelGGL(update64_kernel, dim3(B), dim3(256), spread64(B), s, A, Np, k, k, 1, k - 1, k, B, 1, 0);   // block k-1 into (k,k)
        }
        hipLaunchKernelGGL(diag64_kernel<false>, dim3(B), dim3(256), DIAG64_LDS_BYTES, s, A, Winv, info, Np, k, nullptr);
        if (k + 1 < n) {
            if (k >= 1) VOLT_TRY64(hipStreamWaitEvent(s, ev_c, 0));            // the panel tiles have their last block
            hipLaunchKernelGGL(trsm64_kernel, dim3((n - k - 1) * B), dim3(256), spread64((n - k - 1) * B), s, A, Winv, Np, k, B);
            VOLT_TRY64(hipEventRecord(ev_a, s));                               // column k complete
        }
    }
    // every launch of the auxiliary stream has been waited for by the caller's stream (c before trsm(n-2), b before C(n-1))
    if (Y) {                                                                  // the last row of the inverse, then join
        VOLT_TRY64(hipEventRecord(ev_a, s));
        { const int rc_ = tri_step(n); if (rc_) return rc_; }
        VOLT_TRY64(hipEventRecord(ev_d, ax.aux2));
        VOLT_TRY64(hipStreamWaitEvent(s, ev_d, 0));
    }
    VOLT_LAUNCH_CHECK();
    return 0;
}

extern "C" {

}  // extern "C"
