// The fp64 factorisation (and the triangular inverse of the gradient step) of a SMALL batch in ONE launch (SURVEY 8 row a6 in
// fp64; round 5).
//   reference call sites: psd_safe_cholesky at voltron/rollout_utils.py:35 on the noise-free train block (fp64: condition
//   1e6 .. 1e8), VoltronGP.py:83; a double-precision model's training step (train_utils.py:243-254 in the caller's dtype).
//
// chol64.hip walks a block column as three launches on a chain / bulk multi-stream schedule: 130 us per block column of which
// the diagonal block is 58 -- the rest is launch boundaries and event hand-offs (1 x 4096: 4.1 ms for 32 columns).  Here, as
// in batch_step.hip for fp32, workgroup w runs piece w of a topologically ordered list and the pieces hand their tiles on
// through per-matrix progress words, chasing their inputs one 128-wide K block at a time (common.h, Chase):
//     D(i)      i >= 1: tile (i,i-1) -- A[i,i-1] -= L[i,:i-1] L[i-1,:i-1]^T chasing rows i and i-1 -- parked in an LDS image and
//               solved against L[i-1,i-1] 32 COLUMNS AT A TIME while that block is still being factored (its workgroup
//               publishes every 32-column slice and the slice's inverse as they become final: tiles64.h, diag64_body /
//               trsm64_step); every finished slice goes into the sum of A[i,i] (rank-32, from the image) which then is
//               factored + inverted in the same image (diag64_body) -> L[i,i-1], L[i,i], W_i.
//               Publishes rowp[i] = i, sub[i] = 1 .. 4 (sub-block columns of L[i,i] out), wdone = i + 1
//     LA(i)     i >= 2: A[i,i] -= sum_{m<i-1} L[i,m] L[i,m]^T, chasing row i, parked in place;        publishes la[i] = 1
//     US(i,k)   i >= k + 2: A[i,k] -= L[i,:k] L[k,:k]^T (chasing rows i and k) into the image, the same blocked solve;
//                                                                                            publishes rowp[i] = k + 1
//     TD(i)     Y[i,i] = W_i^T;                                                              publishes tcol[i] = 1
//     T(i,j)    S = Y[j, j..i) L[i, j..i)^T (chasing tcol[j] and rowp[i]), Y[j,i] = -S W_i^T; publishes tcol[j] = i - j + 1
// What a block column costs on the latency chain is then the diagonal block's pivots (~40 us) plus ONE 32-column step and one
// rank-32 update (~7 us): the inverse W_k, the write-out and every hand-off of a whole tile are off it.
// The list needs no table: D(0), then per block column k: LA(k+1), D(k+1) -- its sums under way while D(k) is still at its
// pivots --, US(k+2 .. n-1, k), then row k-1 of the inverse (TD, T longest first); every position for all B matrices with the
// matrix innermost, and a piece is a closed-form function of its index.  The pieces are pulled BY TICKET by a grid of resident
// workgroups (round 6; common.h, "who runs which piece"): every piece waits only for pieces listed before it and a ticket is
// taken by a running workgroup, so whatever is waited for is running or finished -- wherever and in whatever order workgroups
// start.  The diagonal block's image takes 133 KB of LDS: ONE workgroup per CU,
// which is also what the latency chain wants (batch_step.hip: a pivot chain that shares its CU runs 2.5 x slower) -- and why
// this is the schedule of small batches only; from batch64_max tiles per block column on, chol64.hip's bulk kernels (two per
// CU) win.
// No atomics: unlike the K-sliced launches of chol64.hip the result is the same from run to run.
#include "common.h"
#include "tiles64.h"
#include "host.h"
#include "../../include/volt_hip.h"
#include "../../include/volt_hip_tune.h"
#include <algorithm>

namespace volt {

// progress words per matrix (ints): rowp[n] | tcol[n] | sub[n] | la[n] | wdone | tsl[n], padded to a multiple of 32
static inline int batch64_pstride(int n) { return (5 * n + 1 + 31) & ~31; }
static inline int64_t batch64_count(int B, int n, bool has_y) {
    const int64_t per = n + (n >= 3 ? n - 2 : 0) + (int64_t)(n - 1) * (n - 2) / 2 + (has_y ? (int64_t)n * (n + 1) / 2 : 0);
    return (int64_t)B * per;
}

// inv_n > 0 (the inverse alone, volt_trtri_ws_f64): the factor is complete before the launch -- rowp[i] = n, wdone = n (every
// block row of L and every W_i is there), only the inverse's own words (tcol) start at 0
__global__ void batch64_begin_kernel(int* __restrict__ info, int ninfo, int* __restrict__ prog, int nprog, int inv_n, int pstride,
                                     int nwords) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ninfo) info[i] = 0;
    for (int c = i; c < nprog; c += gridDim.x * blockDim.x) {
        int v = 0;
        if (inv_n > 0 && c < nwords) {
            const int o = c % pstride;                        // rowp[n] | tcol[n] | sub[n] | la[n] | wdone | tsl[n]
            if (o < inv_n || o == 4 * inv_n) v = inv_n;
        }
        prog[c] = v;
    }
}

enum Piece64Kind { P64_DIAG = 0, P64_PANEL = 1, P64_TRTRI_DIAG = 2, P64_TRTRI = 3, P64_LOOKAHEAD = 4 };
struct Piece64 { int kind, row, col; };
// position p of a matrix's list (header comment); scalar work, at most n steps
__device__ __host__ inline Piece64 batch64_piece(int p, int n, bool has_y) {
    if (p == 0) return {P64_DIAG, 0, 0};
    p -= 1;
    for (int k = 0; k < n; ++k) {
        if (k >= 1 && k + 1 < n) {
            if (p == 0) return {P64_LOOKAHEAD, k + 1, k + 1};
            p -= 1;
        }
        if (k + 1 < n) {
            if (p == 0) return {P64_DIAG, k + 1, k + 1};
            p -= 1;
        }
        const int nus = n - k - 2 > 0 ? n - k - 2 : 0;
        if (p < nus) return {P64_PANEL, k + 2 + p, k};
        p -= nus;
        if (has_y && k >= 1) {
            if (p == 0) return {P64_TRTRI_DIAG, k - 1, k - 1};
            p -= 1;
            if (p < k - 1) return {P64_TRTRI, k - 1, p};
            p -= k - 1;
        }
    }
    if (p == 0) return {P64_TRTRI_DIAG, n - 1, n - 1};
    return {P64_TRTRI, n - 1, p - 1};
}

// position p of a matrix's list when only the inverse runs: row by row, the diagonal tile first, then the row's tiles longest first
__device__ __host__ inline Piece64 batch64_trtri_piece(int p, int n) {
    int i = 0;
    while (p > i) {                                          // row i has i + 1 pieces
        p -= i + 1;
        ++i;
    }
    if (p == 0) return {P64_TRTRI_DIAG, i, i};
    return {P64_TRTRI, i, p - 1};
}

// the wave's 64x64 of a 128x128 tile of doubles <-> accumulator layout (VOLT_ACC64_RC).  WT: written through at agent scope
// (sc1) -- a tile handed to workgroups on other XCDs leaves nothing behind in this L2 for a release to write back
__device__ __forceinline__ void tile64_load_neg(f64x4 (&v)[16], const double* __restrict__ C, int64_t ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                VOLT_ACC64_RC(mt, nt, q)
                v[mt * 4 + nt][q] = -C[(int64_t)r * ld + c];
            }
}
// element (gi, gj) of the input matrix of series b, from the caller's K (prepare64_kernel's arithmetic)
__device__ __forceinline__ double input64(const KSource64& src, const double* __restrict__ Kb, double add, int gi, int gj) {
    double v = (gi < src.N && gj < src.N) ? Kb[(int64_t)gi * src.ldk + gj] : 0.0;
    if (gi == gj) v = (gi < src.N) ? v + add : 1.0;
    return v;
}
// -(tile (ti, tj) of the input) in the accumulator layout: from the prepared copy A, or straight from K
__device__ __forceinline__ void tile64_input_neg(f64x4 (&v)[16], const KSource64& src, int b, const double* __restrict__ Ab,
                                                 int Np, int ti, int tj) {
    if (!src.K) {
        tile64_load_neg(v, Ab + (int64_t)ti * TS * Np + (int64_t)tj * TS, Np);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double add = (src.sigma2 ? src.sigma2[b] : 0.0) + src.jitter;
    const double* Kb = src.K + (int64_t)b * src.bsk;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                VOLT_ACC64_RC(mt, nt, q)
                v[mt * 4 + nt][q] = -input64(src, Kb, add, ti * TS + r, tj * TS + c);
            }
}
// the same for the diagonal tile's owners (tiles64.h diag_tile)
__device__ __forceinline__ void diag_input_neg(f64x4 (&v)[DIAG_OWN], const KSource64& src, int b, const double* __restrict__ Ab, int Np, int ti) {
    if (!src.K) {
        diag_load_neg(v, Ab + (int64_t)ti * TS * Np + (int64_t)ti * TS, Np);
        return;
    }
    const int lane = threadIdx.x & 63, g = diag_group();
    const double add = (src.sigma2 ? src.sigma2[b] : 0.0) + src.jitter;
    const double* Kb = src.K + (int64_t)b * src.bsk;
#pragma unroll
    for (int t = 0; t < DIAG_OWN; ++t) {
        int tr, tc;
        diag_tile(g, t, tr, tc);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[t][q] = -input64(src, Kb, add, ti * TS + 16 * tr + (lane >> 4) + 4 * q, ti * TS + 16 * tc + (lane & 15));
    }
}
template <bool WT>
__device__ __forceinline__ void tile64_store(const f64x4 (&v)[16], double* __restrict__ C, int64_t ld, double sign) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                VOLT_ACC64_RC(mt, nt, q)
                const double x = sign * v[mt * 4 + nt][q];
                if constexpr (WT) __hip_atomic_store(&C[(int64_t)r * ld + c], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else C[(int64_t)r * ld + c] = x;
            }
}
// -acc into the LDS image (row stride DT64); LOWER: the lower triangle, zeros above it
template <bool LOWER>
__device__ __forceinline__ void tile64_to_image(const f64x4 (&v)[16], double* __restrict__ sT) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                VOLT_ACC64_RC(mt, nt, q)
                sT[r * DT64 + c] = (!LOWER || c <= r) ? -v[mt * 4 + nt][q] : 0.0;
            }
}

// tuning only (volt_tune_batch64_stamps): 8 int64 per workgroup -- s_memrealtime at entry [0] and exit [1], hardware id [2],
// behind the chased sum [3]; progressive tiles: sub-block 3 of the diagonal block seen [4], the tile complete [5]; tiles of
// the inverse: W seen [4], second product done [5]
static long long* g_batch64_stamps = nullptr;

struct Batch64Args {
    double* A; double* Winv; double* Y; int* info;
    int Np, B;
    int* prog;
    int pstride;
    KSource64 src;
    long long* stamps;
    int npieces;
    int inverse_only;        // 1: the factor is there, the launch runs the rows of the inverse alone (volt_trtri_ws_f64)
};
extern __shared__ __attribute__((aligned(16))) double g_sT64[];     // the diagonal block's image / the staging buffers
constexpr int BATCH64_LDS_BYTES = DIAG64_LDS_BYTES + TRSM64_STAGE_BYTES;   // 159 488 of the CU's 163 840
static __shared__ int g_piece64;                                     // the piece thread 0 pulled

// One piece of the list.  LOCAL: the batch is a multiple of 8 -- every piece of a matrix runs under ONE XCD's L2 (the pullers read
// their XCC id and take the matrices of that XCD's queues: common.h) and that L2 is where its tiles are handed on: plain stores
// and a drain, no fences (common.h, LOCALP).
template <bool LOCAL>
__device__ __forceinline__ void batch64_piece(const Batch64Args a, const int w) {
    double* const sT = g_sT64;
    float* smem = reinterpret_cast<float*>(sT);
    double* const A = a.A; double* const Winv = a.Winv; double* const Y = a.Y; int* const info = a.info;
    const int Np = a.Np, B = a.B, pstride = a.pstride;
    int* const prog = a.prog;
    const KSource64& src = a.src;
    long long* const stamps = a.stamps;
    const int n = Np / TS, b = w % B;
    const Piece64 pc = a.inverse_only ? batch64_trtri_piece(w / B, n) : batch64_piece(w / B, n, Y != nullptr);
#define VOLT_B64_STAMP(i) \
    do { if (stamps && threadIdx.x == 0) stamps[(int64_t)w * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#ifdef VOLT_B64_SLICE_STAMPS
#define VOLT_B64_STAMP67(KB)
#else
#define VOLT_B64_STAMP67(KB) if (KB == 0) VOLT_B64_STAMP(6); if (KB == 1) VOLT_B64_STAMP(7);
#endif
    int* rowp = prog + (int64_t)b * pstride;
    int* tcol = rowp + n;
    int* sub = tcol + n;
    int* la = sub + n;
    int* wdone = la + n;
    int* tsl = wdone + 1;                // tsl[i]: 32-column slices of tile (i,i-1) that are final in memory (D(i) raises it 1 .. 4)
    int* info_b = info + b;
    double* Ab = A + (int64_t)b * Np * Np;
    double* Wb = Winv + (int64_t)b * n * TS * TS;
    const int lane = threadIdx.x & 63;
    constexpr int CPB = TS / BK64;
    bool ok = true;

    if (pc.kind == P64_LOOKAHEAD) {
        // A[i,i] -= sum_{m < i-1} L[i,m] L[i,m]^T, chasing row i, parked in place: all of the diagonal tile's sum that does not
        // need block column i-1 -- the tile itself (DIAG below) has the blocked solve of L[i,i-1] in that time
        const int i = pc.row;
        double* C = Ab + (int64_t)i * TS * Np + (int64_t)i * TS;
        f64x4 acc[16];
        tile64_input_neg(acc, src, b, Ab, Np, i, i);
        Chase ch;
        ch.p0 = ch.p1 = rowp + i;
        const double* Li = Ab + (int64_t)i * TS * Np;
        gemm64_nt_128<true, LOCAL, 2>(Li, Np, Li, Np, (i - 1) * CPB, acc, smem, &ch, &ok);
        if (!ok && lane == 0) atomicCAS(info_b, 0, (int)0x80000000);
        tile64_store<!LOCAL>(acc, C, Np, -1.0);
        batch_publish_wt<LOCAL>(la + i, 1);
        return;
    }
    if (pc.kind == P64_DIAG || pc.kind == P64_PANEL) {
        // A tile (i,k) below diagonal block k, solved against L_kk 32 columns at a time as block k's sub-blocks are published
        // (trsm64_step).  DIAG i = k + 1: the tile next to the diagonal, whose row makes the NEXT diagonal block -- the same
        // workgroup adds its slices' products to that block's sum as they become final and then factors it: between the last
        // pivot of block k and the first of block k + 1 there is one 32-column step, one rank-32 update and no hand-off.
        const bool dg = pc.kind == P64_DIAG;
        const int i = pc.row, k = dg ? i - 1 : pc.col;
        if (k >= 0) {
            double* P = Ab + (int64_t)i * TS * Np + (int64_t)k * TS;
            {
                f64x4 acc[16];
                tile64_input_neg(acc, src, b, Ab, Np, i, k);
                if (k > 0) {
                    Chase ch;
                    ch.p0 = rowp + i;
                    ch.p1 = rowp + k;
                    const double *X = Ab + (int64_t)i * TS * Np, *Z = Ab + (int64_t)k * TS * Np;
                    // DIAG: the last K block, L[k,k-1], comes from the tile that has just become block k's diagonal -- it is
                    // always the late one.  A loop of its own: the blocks before it are finished (every staged chunk used)
                    // instead of stopping four chunks short of the wait
                    const int early = dg ? k - 1 : k;
                    gemm64_nt_128<true, LOCAL, 2>(X, Np, Z, Np, early * CPB, acc, smem, &ch, &ok);
                    if (dg) {
                        // X's block: tile (i,k-1), whole.  Z's: D(k)'s own tile, which it hands on SLICE BY SLICE (tsl[k]) -- the
                        // product follows D(k)'s solve 32 columns of K behind instead of starting when the whole tile is out
                        // (14 us of MFMA work on this one CU, all of it on the chain through round 5)
                        ch.p1 = ch.p0;
                        ch.base0 = ch.base1 = k - 1;
                        chase_wait<LOCAL>(ch, 1, ok);
#ifdef VOLT_B64_SLICE_STAMPS
                        VOLT_B64_STAMP(6);
#endif
                        ch.p0 = ch.p1 = tsl + k;
                        // one short loop per slice: the chunk pipeline asks for data four chunks AHEAD of the one it multiplies, so
                        // a single loop over the block would hold slice 0's product back until slice 2 is there
                        for (int sl = 0; sl < 4; ++sl) {
                            ch.base0 = ch.base1 = sl;
#ifdef VOLT_B64_SLICE_STAMPS
                            if (sl == 3) { chase_wait<LOCAL>(ch, 1, ok); VOLT_B64_STAMP(7); }
#endif
                            gemm64_nt_128<true, LOCAL, 2>(X + (int64_t)(k - 1) * TS + 32 * sl, Np, Z + (int64_t)(k - 1) * TS + 32 * sl, Np, CPB / 4, acc, smem, &ch, &ok);
                        }
                    }
                }
                tile64_to_image<false>(acc, sT);          // (the staging buffers are free: the loop ends with a barrier)
            }
            VOLT_B64_STAMP(3);
            const double* Lkk = Ab + (int64_t)k * TS * Np + (int64_t)k * TS;
            const double* Wk = Wb + (int64_t)k * TS * TS;
            double* stage = sT + TS * DT64 + 160;            // three operand blocks of a solve step, behind the image and its pivots
            // KB = 0: a REAL acquire even under LOCAL -- A[k,k] is the one address written twice per launch (LA(k) parks its sum
            // there, from a prepared copy it also READ the input there; D(k) then stores L_kk): a CU that ran LA(k) could hit its
            // old L1 lines when it reads the L_kk sub-blocks below.  One buffer_inv per piece.
            f64x4 accT[DIAG_OWN];
            if (!dg) {
#define VOLT_B64_STEP(KB)                                                       \
                if (KB == 0) batch_wait<false>(sub + k, KB + 1, nullptr, 0, info_b);  \
                else batch_wait<LOCAL>(sub + k, KB + 1, nullptr, 0, info_b);    \
                if (KB == 3) VOLT_B64_STAMP(4);                                 \
                trsm64_step_staged<KB, true>(sT, stage, Lkk, Np, Wk, P, Np);
                VOLT_B64_STEP(0)
                VOLT_B64_STEP(1)
                VOLT_B64_STEP(2)
                VOLT_B64_STEP(3)
#undef VOLT_B64_STEP
            } else {
                // The tile next to the diagonal.  Per step: the slice stays in the image; after the barrier the rank-32 update
                // needs anyway, wave 1 -- the quadrant above the diagonal needs no update -- writes it out (written through
                // unless LOCAL), drains, and raises tsl[i] (and, with the last slice, rowp[i] = k + 1): nothing on the chain
                // waits for stores or for a release.
                // (Measured and not kept: the NEXT step's operands requested before the rank-32 update when block k's next
                // sub-block was already announced, and the diagonal tile's sum fetched behind step 0's operands instead of in
                // front of its wait -- steps 1 .. 3 each 1 us shorter, step 0 5.7 us longer, 1 x 4096 1.80 ms against 1.765.)
                const int wv = threadIdx.x >> 6;
                if (i >= 2) {                                // the diagonal tile's sum so far: the look-ahead's, or the input itself
                    batch_wait<LOCAL>(la + i, 1, nullptr, 0, info_b);
                    if (wv != 1) diag_load_neg(accT, Ab + (int64_t)i * TS * Np + (int64_t)i * TS, Np);
                } else if (wv != 1) {
                    diag_input_neg(accT, src, b, Ab, Np, i);
                }
#define VOLT_B64_DSTEP(KB)                                                      \
                if (KB == 0) batch_wait<false>(sub + k, 1, nullptr, 0, info_b); \
                else batch_wait<LOCAL>(sub + k, KB + 1, nullptr, 0, info_b);    \
                if (KB == 3) VOLT_B64_STAMP(4);                                 \
                VOLT_B64_STAMP67(KB)                                            \
                trsm64_step_staged<KB, false>(sT, stage, Lkk, Np, Wk, P, Np);                 \
                __syncthreads();                                                \
                if (wv == 1) {                                                  \
                    slice64_out<KB, !LOCAL>(sT, P, Np);                         \
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            \
                    if (lane == 0) {                                            \
                        if constexpr (LOCAL) {                                  \
                            *reinterpret_cast<volatile int*>(tsl + i) = KB + 1; \
                            if (KB == 3) *reinterpret_cast<volatile int*>(rowp + i) = k + 1; \
                        } else {                                                \
                            __hip_atomic_store(tsl + i, KB + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                            if (KB == 3) __hip_atomic_store(rowp + i, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                        }                                                       \
                    }                                                           \
                } else {                                                        \
                    diag_syrk_slice<KB>(accT, sT);                              \
                }
                VOLT_B64_DSTEP(0)
                VOLT_B64_DSTEP(1)
                VOLT_B64_DSTEP(2)
                VOLT_B64_DSTEP(3)
#undef VOLT_B64_DSTEP
            }
            if (!ok && lane == 0) atomicCAS(info_b, 0, (int)0x80000000);
            if (!dg) {
                batch_publish_release<LOCAL>(rowp + i, k + 1);
                VOLT_B64_STAMP(5);
                return;
            }
            __syncthreads();                                     // (the last rank-32 update has read the image)
            VOLT_B64_STAMP(5);
            diag_to_image(accT, sT);
        }
        bool image = k >= 0;
        if (!image && src.K) {                           // block 0 straight from K into the image
            const double add = (src.sigma2 ? src.sigma2[b] : 0.0) + src.jitter;
            const double* Kb = src.K + (int64_t)b * src.bsk;
            // eight loads in flight per thread (clamped addresses, the padding and the upper triangle selected afterwards): one
            // element at a time this was 64 round trips in front of the first pivot of the whole factorisation (~15 us)
            const int N = src.N;
#pragma unroll 1
            for (int e0 = threadIdx.x; e0 < TS * TS; e0 += 8 * NT) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * NT, r = e >> 7, c = e & 127;
                    v[u] = Kb[(int64_t)(r < N ? r : N - 1) * src.ldk + (c < N ? c : N - 1)];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * NT, r = e >> 7, c = e & 127;
                    double x = (r < N && c < N) ? v[u] : 0.0;
                    if (r == c) x = (r < N) ? x + add : 1.0;
                    sT[r * DT64 + c] = (c <= r) ? x : 0.0;
                }
            }
            image = true;
        }
#ifdef VOLT_B64_DIAG_STAMPS                              // tuning build: the diagonal block's own 32 stamps (s_memtime) behind the pieces'
        diag64_body<true, LOCAL>(A, Winv, info, Np, i, b, sT, stamps ? stamps + (int64_t)a.npieces * 8 + 32 * (int64_t)i * B : nullptr,
                                 image, sub + i);
#else
        diag64_body<false, LOCAL>(A, Winv, info, Np, i, b, sT, nullptr, image, sub + i);
#endif
        batch_publish_release<LOCAL>(wdone, i + 1);
        return;
    }
    if (pc.kind == P64_TRTRI_DIAG) {
        const int i = pc.row;
        batch_wait<LOCAL>(wdone, i + 1, nullptr, 0, info_b);
        trtri64_diag_body(Wb + (int64_t)i * TS * TS, Y + (int64_t)b * Np * Np + (int64_t)i * TS * Np + (int64_t)i * TS, Np, smem);
        batch_publish_release<LOCAL>(tcol + i, 1);
        return;
    }
    // ---- a tile (i,j) of the inverse: a chased sum, then the product with W_i
    {
        const int i = pc.row, j = pc.col;
        double* Yb = Y + (int64_t)b * Np * Np;
        Chase ch;
        ch.p0 = tcol + j;                                        // tiles (j, j .. i-1) of the inverse: tcol[j] of them are there
        ch.p1 = rowp + i;                                        // L[i, j .. i-1]: block m is there once rowp[i] >= j + m + 1
        ch.base1 = j;
        double* mid = Yb + (int64_t)i * TS * Np + (int64_t)j * TS;       // the unused slot below the diagonal
        f64x4 acc[16];
        zero_acc64(acc);
        gemm64_nt_128<true, LOCAL, 2>(Yb + (int64_t)j * TS * Np + (int64_t)j * TS, Np, Ab + (int64_t)i * TS * Np + (int64_t)j * TS, Np,
                                      (i - j) * CPB, acc, smem, &ch, &ok);
        tile64_store<false>(acc, mid, Np, 1.0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the barrier of the wait below covers the workgroup)
        VOLT_B64_STAMP(3);
        batch_wait<LOCAL>(wdone, i + 1, nullptr, 0, info_b);
        VOLT_B64_STAMP(4);
        zero_acc64(acc);
        gemm64_nt_128<false, false, 2>(mid, Np, Wb + (int64_t)i * TS * TS, TS, CPB, acc, smem);   // (all of `mid` is read before the closing barrier)
        VOLT_B64_STAMP(5);
        if (!ok && lane == 0) atomicCAS(info_b, 0, (int)0x80000000);
        tile64_store<!LOCAL>(acc, Yb + (int64_t)j * TS * Np + (int64_t)i * TS, Np, -1.0);
        batch_publish_wt<LOCAL>(tcol + j, i - j + 1);
    }
#undef VOLT_B64_STAMP
}

template <bool LOCAL>
__global__ __launch_bounds__(256) void batch64_step_kernel(Batch64Args a, int xskew, int xdrop) {
    int* const qw = a.prog + (int64_t)a.B * a.pstride;       // the queue words sit behind the progress words
    const int hw = hw_xcc_id();
    if (LOCAL && ((xdrop >> hw) & 1)) return;
    const int xcc = (hw + xskew) & 7;
    const int per_queue = LOCAL ? a.npieces / 8 : a.npieces;
    BatchPull pull;
    // the pullers' loop, hidden from the loop optimiser by a second (never taken) way in: batch_step.hip
    int w;
    if (xdrop & 0x40000000) {
        w = xskew | (int)0x80000000;
        goto piece;
    }
pull_next:
    __syncthreads();                                         // the last piece's LDS traffic (and its read of g_piece64) is over
    if (threadIdx.x == 0) g_piece64 = batch_next_piece<LOCAL>(pull, qw, xcc, per_queue);
    __syncthreads();
    w = g_piece64;
piece:
    w = __builtin_amdgcn_readfirstlane(w);
    if (w < 0) return;
    if (a.stamps && threadIdx.x == 0) {
        a.stamps[(int64_t)w * 8] = __builtin_amdgcn_s_memrealtime();
        a.stamps[(int64_t)w * 8 + 2] = ((long long)hw << 32) | (unsigned)__builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);
    }
    batch64_piece<LOCAL>(a, w);
    if (a.stamps && threadIdx.x == 0) a.stamps[(int64_t)w * 8 + 1] = __builtin_amdgcn_s_memrealtime();
    goto pull_next;
}

}  // namespace volt

using namespace volt;

// Where the one launch replaces chol64.hip's schedules: the measured crossovers of profiles/r05/batch64_gate_sweep.txt (host.h).
// (B, n, inverse?) only, like every gate: volt_*_workspace_bytes and the step agree on it.
bool volt_internal_batch64_applies(int B, int n, int has_y) {
    const Tunables& tn = tunables();
    if (tn.batch64 <= 0 || B < 1 || n < 2 || B > 65535) return false;
    if (tn.batch64 >= 2) return true;
    const int64_t tiles = (int64_t)B * (n + 1);
    if (has_y) return tiles <= (n >= 24 ? tn.batch64_max_step : tn.batch64_max);
    return tiles <= tn.batch64_max;
}

size_t volt_internal_batch64_bytes(int B, int n, int has_y) {
    if (!volt_internal_batch64_applies(B, n, has_y)) return 0;
    return (((size_t)B * batch64_pstride(n) + BATCH_QWORDS) * sizeof(int) + 255) & ~(size_t)255;   // progress words, then the pullers' queue words
}

// Returns 1 when the step was enqueued, 0 when the shape is not this schedule's (nothing enqueued), a HIP error otherwise.
// src (optional): the caller's K -- the tiles are read from it and A only receives the factor (no copy-in pass).
int volt_internal_batch64_step(double* A, double* Winv, int* info, double* Y, int B, int Np, void* state, size_t state_bytes,
                               void* stream, const KSource64* ksrc) {
    const KSource64 src = ksrc ? *ksrc : KSource64{nullptr, 0, 0, nullptr, 0.0, Np};
    const int n = Np / TS;
    const int has_y = Y != nullptr;
    if (!state || !volt_internal_batch64_applies(B, n, has_y) || state_bytes < volt_internal_batch64_bytes(B, n, has_y)) return 0;
    hipStream_t s = (hipStream_t)stream;
    int* prog = reinterpret_cast<int*>(state);
    const int pstride = batch64_pstride(n), nprog = B * pstride + BATCH_QWORDS;
    int blocks = (std::max(B, nprog) + 255) / 256;
    if (blocks > 256) blocks = 256;
    if (blocks * 256 < B) blocks = (B + 255) / 256;
    hipLaunchKernelGGL(batch64_begin_kernel, dim3(blocks), dim3(256), 0, s, info, B, prog, nprog, 0, pstride, B * pstride);
    // eight queues, one per XCD, when the matrices divide among them evenly (the pullers read their XCC id: common.h)
    const bool local = (B & 7) == 0 && tunables().batch_local != 0 && tunables().xccs == 8;
    const int64_t npieces = batch64_count(B, n, Y != nullptr);
    if (npieces > 0x7fffffff) return 0;
    // as many pullers as the chip holds at once: one per CU (the image's LDS); nothing depends on the number
    const unsigned grid = (unsigned)std::min<int64_t>(npieces, tunables().batch_pullers > 0 ? (int64_t)tunables().cus * tunables().batch_pullers : npieces);
    const Batch64Args args{A, Winv, Y, info, Np, B, prog, pstride, src, g_batch64_stamps, (int)npieces, 0};
    const int xskew = tunables().batch_xskew, xdrop = tunables().batch_xdrop;
    hipError_t e;
    if (local) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(batch64_step_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, BATCH64_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(batch64_step_kernel<true>, dim3(grid), dim3(256), BATCH64_LDS_BYTES, s, args, xskew, xdrop);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(batch64_step_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, BATCH64_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(batch64_step_kernel<false>, dim3(grid), dim3(256), BATCH64_LDS_BYTES, s, args, xskew, xdrop);
    }
    e = hipGetLastError();
    return e != hipSuccess ? (int)e : 1;
}

// The triangular inverse alone as one launch: the rows of the inverse are pieces of the one-launch step already
// (TD / T above: a chased sum, then the product with W_i); with the factor complete they only wait for each other -- tile (i,j)
// for tiles (j .. i-1, j) of its column, one K block behind.  chol64.hip's volt_trtri_f64 walks the rows as launches (two
// kernels per row, a look-ahead stream): 8 x 4096 4.8 ms; here 528 tiles per matrix keep every CU busy from the first row on.
// The error word of a hand-off time-out goes to `ierr` [B] in the caller's scratch (the inverse has no info argument).
// Returns 1 when enqueued, 0 when the shape is not this schedule's, a HIP error otherwise.
size_t volt_internal_batch64_trtri_bytes(int B, int n) {
    const Tunables& tn = tunables();
    if (tn.batch64 <= 0 || B < 1 || n < 2 || B > 65535) return 0;
    if (tn.batch64 < 2 && (int64_t)B * (n + 1) > tn.batch64_max) return 0;
    return ((((size_t)B * batch64_pstride(n) + BATCH_QWORDS) * sizeof(int) + 255) & ~(size_t)255) + (((size_t)B * sizeof(int) + 255) & ~(size_t)255);
}

int volt_internal_batch64_trtri(const double* A, const double* Winv, double* Y, int B, int Np, void* state, size_t state_bytes,
                                void* stream) {
    const int n = Np / TS;
    const size_t need = volt_internal_batch64_trtri_bytes(B, n);
    if (!state || !need || state_bytes < need) return 0;
    hipStream_t s = (hipStream_t)stream;
    int* prog = reinterpret_cast<int*>(state);
    const int pstride = batch64_pstride(n), nprog = B * pstride + BATCH_QWORDS;
    int* ierr = reinterpret_cast<int*>(reinterpret_cast<char*>(state) + ((((size_t)nprog) * sizeof(int) + 255) & ~(size_t)255));
    int blocks = (std::max(B, nprog) + 255) / 256;
    if (blocks > 256) blocks = 256;
    if (blocks * 256 < B) blocks = (B + 255) / 256;
    hipLaunchKernelGGL(batch64_begin_kernel, dim3(blocks), dim3(256), 0, s, ierr, B, prog, nprog, n, pstride, B * pstride);
    const bool local = (B & 7) == 0 && tunables().batch_local != 0 && tunables().xccs == 8;
    const int64_t npieces = (int64_t)B * n * (n + 1) / 2;
    if (npieces > 0x7fffffff) return 0;
    const unsigned grid = (unsigned)std::min<int64_t>(npieces, tunables().batch_pullers > 0 ? (int64_t)tunables().cus * tunables().batch_pullers : npieces);
    const KSource64 src{nullptr, 0, 0, nullptr, 0.0, Np};
    const Batch64Args args{const_cast<double*>(A), const_cast<double*>(Winv), Y, ierr, Np, B, prog, pstride, src, g_batch64_stamps, (int)npieces, 1};
    const int xskew = tunables().batch_xskew, xdrop = tunables().batch_xdrop;
    hipError_t e;
    if (local) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(batch64_step_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, BATCH64_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(batch64_step_kernel<true>, dim3(grid), dim3(256), BATCH64_LDS_BYTES, s, args, xskew, xdrop);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(batch64_step_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, BATCH64_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(batch64_step_kernel<false>, dim3(grid), dim3(256), BATCH64_LDS_BYTES, s, args, xskew, xdrop);
    }
    e = hipGetLastError();
    return e != hipSuccess ? (int)e : 1;
}

extern "C" {

int volt_tune_batch64_stamps(long long* stamps) {
    g_batch64_stamps = stamps;
    return 0;
}

// Host only (no GPU): the piece list of the fp64 one-launch step, in grid order -- items [max_items][4] int32
// {kind, row, col, matrix} (kinds: 0 D, 1 US, 2 TD, 3 T).  Returns the number of pieces, -1 bad argument, -2 max_items too small.
int volt_batch64_describe(int B, int n, int has_y, int* items, int max_items) {
    if (B < 1 || n < 1) return -1;
    const int64_t cnt = batch64_count(B, n, has_y != 0);
    if (cnt > 0x7fffffff) return -1;
    if (items) {
        if (cnt > max_items) return -2;
        for (int64_t w = 0; w < cnt; ++w) {
            const Piece64 pc = batch64_piece((int)(w / B), n, has_y != 0);
            items[4 * w] = pc.kind;
            items[4 * w + 1] = pc.row;
            items[4 * w + 2] = pc.col;
            items[4 * w + 3] = (int)(w % B);
        }
    }
    return (int)cnt;
}

}  // extern "C"
