// Device side of the fp64 factorisation (chol64.hip's launch-per-column kernels and batch64_step.hip's one launch call the
// same bodies): the 128x128 fp64 MFMA core with its chunk pipeline, and the diagonal block (factor + inverse in an LDS image).
#pragma once
#include "common.h"

namespace volt {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_64 __attribute__((ext_vector_type(4)));

// Where the one-launch step reads a tile of its input from (batch64_step.hip): K == nullptr -- the prepared copy A; else the
// caller's K (+ sigma2[b] + jitter on the diagonal, identity in the padding): no copy-in pass ahead of the factorisation.
struct KSource64 {
    const double* K;
    int64_t ldk, bsk;
    const double* sigma2;
    double jitter;
    int N;
};

constexpr int SLD64 = SLD / 2;          // 18 doubles per LDS row
constexpr int BK64 = BK / 2;            // 16 doubles of K per chunk

// Fragments of one K step (4 doubles of K) of the staged chunk: lane l supplies A[row = l & 15][k = l >> 4] and
// B[col = l & 15][k = l >> 4] for the four 16-row / 16-column blocks of the wave's 64x64.
struct Frag64 { double a[4], b[4]; };
__device__ __forceinline__ void frag64_load(Frag64& f, const float* __restrict__ buf, int kk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lk = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const double* sA = reinterpret_cast<const double*>(buf);
    const double* sB = reinterpret_cast<const double*>(buf + TS * SLD);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f.a[t] = sA[(wr * 64 + t * 16 + l15) * SLD64 + kk * 4 + lk];
        f.b[t] = sB[(wc * 64 + t * 16 + l15) * SLD64 + kk * 4 + lk];
    }
}
// accumulator register q of lane l of block (mt, nt) is element (row = 16 mt + (l >> 4) + 4 q, col = 16 nt + (l & 15))
__device__ __forceinline__ void frag64_mma_row(const Frag64& f, int mt, f64x4 (&acc)[16]) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
        acc[mt * 4 + nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[mt], f.b[nt], acc[mt * 4 + nt], 0, 0, 0);
}
__device__ __forceinline__ void frag64_mma(const Frag64& f, f64x4 (&acc)[16]) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) frag64_mma_row(f, mt, acc);
}

// One chunk (4 K steps of 4 doubles) of the fp64 K loop, same shape as chunk_run (common.h): fragments
// double-buffered in registers, the order pinned, staging two instructions at a time between groups of four MFMAs.
template <bool STEADY>
__device__ __forceinline__ void chunk64_run(const float* cur, float* nxt, bool more_, Frag64& F0, Frag64& F1, f64x4 (&acc)[16],
                                            StageRegs& s, bool do_st_, bool do_ld_, const StageAddr& sa, int k_ld) {
    const bool more = STEADY || more_, do_st = STEADY || do_st_, do_ld = STEADY || do_ld_;
    frag64_load(F1, cur, 1);
    VOLT_SB();
    frag64_mma(F0, acc);
    VOLT_SB();
    frag64_load(F0, cur, 2);
    VOLT_SB();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        frag64_mma_row(F1, m, acc);
        if (do_st) stage_store_piece(s, nxt, m);
        VOLT_SB();
    }
    frag64_load(F1, cur, 3);
    VOLT_SB();
    const int so = __builtin_amdgcn_readfirstlane(k_ld * 4);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        frag64_mma_row(F0, m, acc);
        if (do_ld) stage_load_piece(s, sa, so, m);
        VOLT_SB();
    }
    __syncthreads();
    if (more) frag64_load(F0, nxt, 0);
    VOLT_SB();
    frag64_mma(F1, acc);
    VOLT_SB();
}

// acc += A[0:128, 0:16 nchunks] * B[0:128, 0:16 nchunks]^T (doubles; lda / ldb in doubles).  The software pipeline of
// gemm_nt_128 on the float view of the operands: loads two chunks ahead, double-buffered LDS, one barrier per chunk.
// CHASE (batch64_step.hip): the operands' 128-wide K blocks are still being produced by other workgroups of the launch; the
// loop asks the two progress words (common.h, Chase) before it requests a chunk of a block it has not yet seen complete.
// SETS: staging register sets = chunks the global loads run ahead of the LDS stores.
//   1  (chol64.hip's kernels, two workgroups per CU): with two sets a wave needs 242 + 128 registers and a SIMD holds one wave;
//      one set fits two, whose barrier and LDS waits cover each other.
//   2  (batch64_step.hip, ONE workgroup per CU -- the diagonal block's image -- so a wave has the 512 registers anyway): the
//      loads are two chunks ahead, as in the fp32 loop.
template <bool CHASE = false, bool LOCALP = false, int SETS = 1>
__device__ __forceinline__ void gemm64_nt_128(const double* __restrict__ A, int64_t lda, const double* __restrict__ B,
                                              int64_t ldb, int nchunks, f64x4 (&acc)[16], float* smem,
                                              const Chase* ch = nullptr, bool* ch_ok = nullptr) {
    if (nchunks <= 0) return;
    constexpr int CPB = TS / BK64;             // chunks per 128-wide K block
    int ready = 0;
    if constexpr (CHASE) ready = chase_wait<LOCALP>(*ch, 1, *ch_ok);
    // chunk j may be requested once block j / CPB is complete (clamped: nothing past the range is asked for)
    auto ask = [&](int j) {
        if constexpr (CHASE) {
            const int need = (j < nchunks ? j : nchunks - 1) / CPB + 1;
            if (ready < need) ready = chase_wait<LOCALP>(*ch, need, *ch_ok);
        }
    };
    const StageAddr sa = stage_addr(reinterpret_cast<const float*>(A), 2 * lda, reinterpret_cast<const float*>(B), 2 * ldb);
    Frag64 F0, F1;
    float* b0 = smem;
    float* b1 = smem + STAGE_FLOATS;
    int c = 0;
    if constexpr (SETS == 1) {
        StageRegs s0;
        stage_load_buf(s0, sa, 0);
        stage_store(s0, smem);
        if (nchunks > 1) stage_load_buf(s0, sa, BK);
        __syncthreads();
        frag64_load(F0, smem, 0);
        for (; c + 3 < nchunks; c += 2) {
            ask(c + 3);
            chunk64_run<true>(b0, b1, true, F0, F1, acc, s0, true, true, sa, (c + 2) * BK);
            chunk64_run<true>(b1, b0, true, F0, F1, acc, s0, true, true, sa, (c + 3) * BK);
        }
        for (; c + 1 < nchunks; c += 2) {
            ask(c + 3);
            chunk64_run<false>(b0, b1, true, F0, F1, acc, s0, true, c + 2 < nchunks, sa, (c + 2) * BK);
            chunk64_run<false>(b1, b0, c + 2 < nchunks, F0, F1, acc, s0, c + 2 < nchunks, c + 3 < nchunks, sa, (c + 3) * BK);
        }
        if (c < nchunks) chunk64_run<false>(b0, b1, false, F0, F1, acc, s0, false, false, sa, 0);
    } else {
        // chunk c stores the set that holds chunk c + 1 and refills it with chunk c + 3; the sets alternate
        StageRegs s0, s1;
        stage_load_buf(s0, sa, 0);
        stage_store(s0, smem);
        if (nchunks > 1) stage_load_buf(s0, sa, BK);
        if (nchunks > 2) stage_load_buf(s1, sa, 2 * BK);
        __syncthreads();
        frag64_load(F0, smem, 0);
        for (; c + 4 < nchunks; c += 2) {
            ask(c + 4);
            chunk64_run<true>(b0, b1, true, F0, F1, acc, s0, true, true, sa, (c + 3) * BK);
            chunk64_run<true>(b1, b0, true, F0, F1, acc, s1, true, true, sa, (c + 4) * BK);
        }
        for (; c + 1 < nchunks; c += 2) {
            ask(c + 4);
            chunk64_run<false>(b0, b1, true, F0, F1, acc, s0, true, c + 3 < nchunks, sa, (c + 3) * BK);
            chunk64_run<false>(b1, b0, c + 2 < nchunks, F0, F1, acc, s1, c + 2 < nchunks, c + 4 < nchunks, sa, (c + 4) * BK);
        }
        if (c < nchunks) chunk64_run<false>(b0, b1, false, F0, F1, acc, s0, false, false, sa, 0);
    }
    __syncthreads();
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

// ----------------------------------------------------------------------------- P2
// One workgroup per matrix factors the 128x128 diagonal block and inverts it, blocked by 32 in an LDS image (row
// stride 129 doubles) -- the structure of the fp32 diag_body (chol.hip):
//   chol32   the 32x32 diagonal sub-block, all 256 threads: thread (ty, tx) keeps the 2x2 cyclic elements
//            (ty + 16 a, tx + 16 c) in registers; per pivot the owners publish the still unscaled column to a
//            double-buffered LDS vector (ONE barrier per pivot), everyone applies a_rc -= a_rj a_cj / d_j
//   inv32    X = L_kk^-1 by forward substitution, one column per lane of one wave, fully unrolled; X replaces L_kk
//            in the image (the panel solve, the trailing updates and the blocked inverse only ever need X)
//   panel    L[i,kb] = A[i,kb] X_kb^T and trailing updates A[i,j] -= L[i,kb] L[j,kb]^T: 32x32x32 products on
//            v_mfma_f64_16x16x4_f64 straight from the image, one wave per block
//   W = L^-1 blocked: W[i,j] = -X_i sum_{m=j}^{i-1} L[i,m] W[m,j], wave j owns block column j; the W[m,j] it
//            produced stay in its accumulators and are fed back as MFMA B operands FROM REGISTERS (accumulator
//            register q of lane l holds row (l >> 4) + 4 q of a 16-row tile: exactly the k index step q wants from
//            that lane)
// Round-2 first version (unblocked LDS loops): 650 us per block; this one: see DESIGN 4.8.
constexpr int DT64 = TS + 1;
constexpr int DIAG64_LDS_BYTES = (TS * DT64 + 128 + 32) * 8;

// acc[tr*2+tc] (16x16 tile at rows 16 tr, cols 16 tc of a 32x32 block) += sign * A[32x32] * B[32x32]^T, A and B row-major
// blocks of the image:  C[r][c] = sum_p A[r][p] B[c][p]
template <bool NEG>
__device__ __forceinline__ void mm64_nt(f64x4 (&acc)[4], const double* __restrict__ A, const double* __restrict__ B) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        double a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            a[t] = A[(t * 16 + l15) * DT64 + 4 * st + lk];
            b[t] = B[(t * 16 + l15) * DT64 + 4 * st + lk];
            if (NEG) a[t] = -a[t];
        }
#pragma unroll
        for (int tr = 0; tr < 2; ++tr)
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
                acc[tr * 2 + tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tr], b[tc], acc[tr * 2 + tc], 0, 0, 0);
    }
}
// acc += A[32x32] (image block, row-major) * Breg, Breg = a 32x32 block held as 4 accumulator tiles [tp*2+tc]
__device__ __forceinline__ void mm64_lds_reg(f64x4 (&acc)[4], const double* __restrict__ A, const f64x4 (&Breg)[4]) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double a[2];
#pragma unroll
            for (int tr = 0; tr < 2; ++tr) a[tr] = A[(tr * 16 + l15) * DT64 + tp * 16 + 4 * q + lk];
#pragma unroll
            for (int tr = 0; tr < 2; ++tr)
#pragma unroll
                for (int tc = 0; tc < 2; ++tc)
                    acc[tr * 2 + tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tr], Breg[tp * 2 + tc][q], acc[tr * 2 + tc], 0, 0, 0);
        }
}
// acc += A[32x32] * B[32x32] (both image blocks, row-major):  C[r][c] = sum_p A[r][p] B[p][c]
__device__ __forceinline__ void mm64_nn(f64x4 (&acc)[4], const double* __restrict__ A, const double* __restrict__ B) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        double a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            a[t] = A[(t * 16 + l15) * DT64 + 4 * st + lk];
            b[t] = B[(4 * st + lk) * DT64 + t * 16 + l15];
        }
#pragma unroll
        for (int tr = 0; tr < 2; ++tr)
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
                acc[tr * 2 + tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tr], b[tc], acc[tr * 2 + tc], 0, 0, 0);
    }
}
__device__ __forceinline__ void acc64_zero(f64x4 (&acc)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = 0.0;
}
// element (r, c) of accumulator register q of tile (tr, tc) inside a 32x32 block
#define VOLT_BLK64_RC(tr, tc, q)                                  \
    const int r = (tr) * 16 + ((threadIdx.x & 63) >> 4) + 4 * (q); \
    const int c = (tc) * 16 + (threadIdx.x & 15);
__device__ __forceinline__ void acc64_load(f64x4 (&acc)[4], const double* __restrict__ C) {
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                VOLT_BLK64_RC(tr, tc, q)
                acc[tr * 2 + tc][q] = C[r * DT64 + c];
            }
}
__device__ __forceinline__ void acc64_store(const f64x4 (&acc)[4], double* __restrict__ C, double sign) {
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                VOLT_BLK64_RC(tr, tc, q)
                C[r * DT64 + c] = sign * acc[tr * 2 + tc][q];
            }
}

// broadcast of a double from one lane: two v_readlane_b32 into an SGPR pair (no LDS round trip, no barrier)
__device__ __forceinline__ double rl64(double v, int src) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// c = fma(-l, broadcast(l, lane l0), c): the SGPR broadcast (two v_readlane_b32 into the FIXED pair s[90:91], declared clobbered)
// and its FMA pinned together in one asm block.  A VALU result needs a wait state before v_readlane reads it, and a VALU may read
// a readlane's SGPR two wait states after it: the s_nops.
__device__ __forceinline__ void rl_fma64_1(double& c0, double l, int lo, int hi, int l0) {
    asm volatile("s_nop 0\n\tv_readlane_b32 s90, %2, %4\n\tv_readlane_b32 s91, %3, %4\n\ts_nop 1\n\t"
                 "v_fma_f64 %0, -%1, s[90:91], %0"
                 : "+v"(c0)
                 : "v"(l), "v"(lo), "v"(hi), "i"(l0)
                 : "s90", "s91");
}

// a[c] += rep[16 R + n] * m in every row R of 16 lanes: the multiplicand comes through the DPP row_newbcast path (the only DPP
// control the 64-bit VALU has), one 8-byte instruction where the SGPR broadcast needs two v_readlane_b32 and a v_fma_f64
__device__ __forceinline__ void fmac64_row(double& c, double rep, double m, int n) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(rep), "v"(m), "i"(n));
}
// lanes src0 + (lane & 15) of v, repeated in all four rows of 16 lanes
__device__ __forceinline__ double rows_of64(int lo, int hi, int src0) {
    const int at = 4 * (src0 + (threadIdx.x & 15));
    const unsigned r0 = (unsigned)__builtin_amdgcn_ds_bpermute(at, lo), r1 = (unsigned)__builtin_amdgcn_ds_bpermute(at, hi);
    return __builtin_bit_cast(double, ((unsigned long long)r1 << 32) | r0);
}

// One WAVE factors the 32x32 diagonal sub-block (kb,kb) with one matrix row per lane in registers -- the idiom of the
// fp32 diagonal block (chol.hip): the 32 dependent pivots cost no barrier and no LDS round trip.  Lanes 0..31 hold the rows of
// the diagonal sub-block; lanes 32..63 the rows of the panel block (prow,kb) below it, which the very same instructions turn
// into L[prow,kb] = A[prow,kb] L_kk^-T -- no inverse is needed on the way down.  Several waves run this side by side, each
// with its own copy of the (tiny) diagonal factorisation and its own panel block; the one with `own` writes L_kk and
// the reciprocal pivots back.
//
// What the phase costs is the NUMBER of instructions: a lone wave issues one 4-byte VALU instruction per 4 clocks and one
// 8-byte instruction per 5 (scripts/ubench/issue_rate.hip), and the pivot chain itself hides under the updates
// (scripts/ubench/phase64.hip takes the phase apart).  Through round 5 every update a[c] -= l_r l_c was two v_readlane_b32
// (l_c into an SGPR pair) and one v_fma_f64, and every pivot carried a wave-uniform branch (d > 0 ?) and an exec-masked LDS
// store of the reciprocal pivot: 5.9 us per 32 pivots, of which the two branches 1.4.  Now:
//   * lane j KEEPS 1 / L[j][j] (two v_cndmask); it is stored, and tested, once after the loop -- a pivot that is not > 0 (or NaN)
//     leaves a NaN there (rsq(0) = inf times 0 in the Newton step, rsq(d < 0) = NaN) and only LATER pivots inherit it, so the
//     first lane with !(rs > 0) is the pivot LAPACK reports;
//   * the l column of a pivot is repeated in all four rows of 16 lanes by ds_bpermute (lanes 0..15's and lanes 16..31's: two
//     copies), and the updates read it through DPP row_newbcast: ONE v_fmac_f64 per element.  Only the next pivot's own column
//     still takes the SGPR path, straight away; the others run one pivot LATE, under the flight of the next bpermute pair, so
//     the LDS crossbar's latency is never waited for;
//   * the products are the same and so is their order per element: the factor is bit-identical to round 5's.
// 3.6 us per 32 pivots alone, 3.8 with a second wave beside it (ubench); in the kernel: scripts/tune_diag64.py.
// (Round 5 tried the broadcast through LDS memory instead -- every lane leaves its l in a per-wave buffer, all lanes read entries
// j+1 .. 31 back 16 bytes at a time -- and measured 6.4 us against 6.9 alone, but 9.5 against 6.3 while another wave inverts the
// previous sub-block from the same LDS.  The first version -- 256 threads, 2x2 cyclic elements each, one barrier per pivot --
// took 19.7 us per sub-block, 79 of the kernel's 139 us.)
__device__ __forceinline__ void pivot_phase64(double* __restrict__ sT, double* __restrict__ rdiag, int kb, int prow, bool own,
                                              int& bad) {
    const int lane = threadIdx.x & 63, l31 = lane & 31;
    const bool up = lane >= 32;
    double* rowp = up ? sT + (32 * (prow < 0 ? kb : prow) + l31) * DT64 + 32 * kb : sT + (32 * kb + l31) * DT64 + 32 * kb;
    const bool live = !up || prow >= 0;
    double a[32], rsv = 1.0;
#pragma unroll
    for (int c = 0; c < 32; ++c) a[c] = live ? rowp[c] : 0.0;
    double rep0p = 0.0, rep1p = 0.0, lp = 0.0;                      // pivot j-1: -l of lanes 0..15 / 16..31 in every row, and l
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const double d = rl64(a[j], j);                             // pivot: row j of the diagonal half
        double rs = __builtin_amdgcn_rsq(d);                        // 1/sqrt(d): v_rsq_f64 + two Newton steps
        rs = rs * (1.5 - 0.5 * d * rs * rs);
        rs = rs * (1.5 - 0.5 * d * rs * rs);
        const double l = a[j] * rs;                                 // lane r: L[r][j]  (lane j: d rs = sqrt d)
        a[j] = l;
        rsv = (lane == j) ? rs : rsv;                               // 1 / L[j][j] for the inverse, and the test below
        const unsigned long long lu = __builtin_bit_cast(unsigned long long, l);
        const int llo = (int)(unsigned)lu, lhi = (int)(unsigned)(lu >> 32);
        double rep0 = 0.0, rep1 = 0.0;                              // -l: the sign goes in with the high word
        if (j + 2 < 16) rep0 = rows_of64(llo, lhi ^ (int)0x80000000u, 0);
        if (j + 2 < 32) rep1 = rows_of64(llo, lhi ^ (int)0x80000000u, 16);
        // the next pivot's column: what pivot j-1 still owes it, then pivot j's own term by the SGPR path
        if (j >= 1 && j + 1 < 32) fmac64_row(a[j + 1], j + 1 < 16 ? rep0p : rep1p, lp, (j + 1) & 15);
        if (j + 1 < 32) rl_fma64_1(a[j + 1], l, llo, lhi, j + 1);
        // pivot j-1's other columns: its bpermutes landed a whole pivot ago
        if (j >= 1) {
#pragma unroll
            for (int c = j + 2; c < 32; ++c) fmac64_row(a[c], c < 16 ? rep0p : rep1p, lp, c & 15);
        }
        rep0p = rep0; rep1p = rep1; lp = l;
    }
    {
        const unsigned nb = (unsigned)__builtin_amdgcn_ballot_w64(!(rsv > 0.0));     // lanes 0..31: the pivots
        if (nb != 0 && bad == 0) bad = 32 * kb + __builtin_ctz(nb) + 1;              // wave-uniform
        if (own && !up) rdiag[32 * kb + l31] = rsv;
    }
    if (up) {
        if (prow >= 0) {
#pragma unroll
            for (int c = 0; c < 32; ++c) rowp[c] = a[c];
        }
    } else if (own) {
#pragma unroll
        for (int c = 0; c < 32; ++c) rowp[c] = (c <= l31) ? a[c] : 0.0;
    }
}

// X = L^-1 for the 32x32 lower block at (32 kb, 32 kb) by ONE wave; X replaces L in the image.  rdiag holds the reciprocal
// pivots.  Call with the block complete in LDS; ends WITHOUT a barrier.
// Lane c (and its twin c + 32) solves L x = e_c column-oriented: x[m] = acc[m] / L[m][m], then acc[r] -= L[r][m] x[m] for every
// r > m -- independent FMAs.  L[r][m] has to reach every lane: the wave keeps -L in registers, rows 0..15 and rows 16..31 each
// repeated in all four rows of 16 lanes (lane l holds rows l & 15 and 16 + (l & 15)), and the FMA reads row r's register through
// DPP row_newbcast -- one instruction per term, no LDS access inside the solve.  (Through round 5 every term was an LDS broadcast
// read and an FMA, row by row; the compiler waited for each read in turn -- 4.6 us in the launch-per-column kernel, 9.1 us in the
// one-launch step where it had become what the pivot phases waited for: scripts/batch64_stamps.py.)
__device__ __forceinline__ void inv32_f64(double* __restrict__ sT, const double* __restrict__ rdiag, int kb) {
    double* D = sT + (32 * kb) * DT64 + 32 * kb;
    const int lane = threadIdx.x & 63, i = lane & 15, c = lane & 31;
    double a0[16], a1[32], x[32];
#pragma unroll
    for (int m = 0; m < 16; ++m) a0[m] = -D[i * DT64 + m];
#pragma unroll
    for (int m = 0; m < 32; ++m) a1[m] = -D[(16 + i) * DT64 + m];
    const double r0 = rdiag[32 * kb + i], r1 = rdiag[32 * kb + 16 + i];
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        double xm = 0.0;
        fmac64_row(xm, m < 16 ? r0 : r1, x[m], m & 15);                  // x[m] / L[m][m]
        x[m] = xm;
#pragma unroll
        for (int r = m + 1; r < 32; ++r) fmac64_row(x[r], r < 16 ? a0[m & 15] : a1[m], xm, r & 15);
    }
    if (lane < 32) {                       // (one exec change, after the last DPP instruction)
#pragma unroll
        for (int r = 0; r < 32; ++r) D[r * DT64 + c] = x[r];
    }
}

// ---- the blocked solve  tile <- tile L_kk^-T  in an LDS image, 32 columns at a time (batch64_step.hip) ---------------------
// B operand of a 32x32x32 product straight from memory into registers: element [c][p] of a row-major block (ld doubles)
// (Fetching these blocks with agent-scope (sc1) loads, so that the wait in front of them needs no acquire fence, was measured:
// the loads are served by memory instead of the L2 and the block column takes 67.6 us instead of 61.2.)
struct Blk64 { double v[8][2]; };
__device__ __forceinline__ void blk64_load(Blk64& b, const double* __restrict__ B, int64_t ld) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int st = 0; st < 8; ++st)
#pragma unroll
        for (int t = 0; t < 2; ++t) b.v[st][t] = B[(int64_t)(t * 16 + l15) * ld + 4 * st + lk];
}
// acc += sign * A[32x32] (image block) * B^T, B in registers:  C[r][c] = sum_p A[r][p] B[c][p]
template <bool NEG>
__device__ __forceinline__ void mm64_nt_rb(f64x4 (&acc)[4], const double* __restrict__ A, const Blk64& b) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        double a[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            a[t] = A[(t * 16 + l15) * DT64 + 4 * st + lk];
            if (NEG) a[t] = -a[t];
        }
#pragma unroll
        for (int tr = 0; tr < 2; ++tr)
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
                acc[tr * 2 + tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tr], b.v[st][tc], acc[tr * 2 + tc], 0, 0, 0);
    }
}
// Step KB on the 128x128 tile P in the image: column slice KB becomes final, Lt = P[:,KB] X_KB^T, and is taken out of the
// slices to its right, P[:,j] -= Lt L[j,KB]^T.  Wave w owns rows 32 w .. 32 w + 31 (no barrier inside, none needed between
// steps).  Lkk: the diagonal block of L in memory (ld doubles), Wk: W_k, whose diagonal 32-blocks are the X_kb.  Then the
// final slice goes out to `out` (ldo doubles): 16-byte stores -- unless the caller has one wave do that for all rows (slice64_out).
struct Trsm64Ops { Blk64 bx, bl[3]; };          // X_KB and the blocks L[j,KB], j > KB, of one step (what a step does not use is dead)
template <int KB>
__device__ __forceinline__ void trsm64_load(Trsm64Ops& o, const double* __restrict__ Lkk, int64_t ld, const double* __restrict__ Wk) {
    blk64_load(o.bx, Wk + (int64_t)(32 * KB) * TS + 32 * KB, TS);
#pragma unroll
    for (int j = KB + 1; j <= 3; ++j) blk64_load(o.bl[j - KB - 1], Lkk + (int64_t)(32 * j) * ld + 32 * KB, ld);
}
template <int KB, bool STORE = true>
__device__ __forceinline__ void trsm64_apply(double* __restrict__ sT, const Trsm64Ops& o, double* __restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* rows = sT + (32 * wave) * DT64;
    f64x4 acc[4];
    acc64_zero(acc);
    mm64_nt_rb<false>(acc, rows + 32 * KB, o.bx);
    acc64_store(acc, rows + 32 * KB, 1.0);
#pragma unroll
    for (int j = KB + 1; j <= 3; ++j) {
        f64x4 c[4];
        acc64_load(c, rows + 32 * j);
        mm64_nt_rb<true>(c, rows + 32 * KB, o.bl[j - KB - 1]);
        acc64_store(c, rows + 32 * j, 1.0);
    }
    if constexpr (STORE) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = it * 4 + (lane >> 4), c = (lane & 15) * 2;
            f64x2 v;
            v[0] = rows[r * DT64 + 32 * KB + c];
            v[1] = rows[r * DT64 + 32 * KB + c + 1];
            *reinterpret_cast<f64x2*>(out + (int64_t)(32 * wave + r) * ldo + 32 * KB + c) = v;
        }
    }
}
template <int KB, bool STORE = true>
__device__ __forceinline__ void trsm64_step(double* __restrict__ sT, const double* __restrict__ Lkk, int64_t ld,
                                            const double* __restrict__ Wk, double* __restrict__ out, int64_t ldo) {
    Trsm64Ops o;
    trsm64_load<KB>(o, Lkk, ld, Wk);
    trsm64_apply<KB, STORE>(sT, o, out, ldo);
}
// The same step with its operand blocks fetched ONCE per workgroup (batch64_step.hip): the four waves need the same <= 4 blocks
// of 8 KB, and blk64_load has every wave pull them with 16 scattered 8-byte loads per lane and block (2.4 us from the wait to
// the first product, stamped).  Here the 256 threads fetch X_KB and the first two L[j,KB] with 16-byte loads (two per thread and
// block) into a stage behind the image (row stride 34 doubles: the fragment reads of a half-wave touch 64 different banks), one
// barrier, and each wave takes its fragments from LDS.  Step 0's fourth block -- the one its products need last -- still
// comes the old way, under the first two products.
constexpr int OPS_LD = 34;
constexpr int OPS_BLOCK = 32 * OPS_LD;                        // doubles per staged block
constexpr int TRSM64_STAGE_BYTES = 3 * OPS_BLOCK * 8;
__device__ __forceinline__ void blk64_from_stage(Blk64& b, const double* __restrict__ st) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
        for (int t = 0; t < 2; ++t) b.v[s8][t] = st[(t * 16 + l15) * OPS_LD + 4 * s8 + lk];
}
template <int KB, bool STORE = true>
__device__ __forceinline__ void trsm64_step_staged(double* __restrict__ sT, double* __restrict__ stage, const double* __restrict__ Lkk,
                                                   int64_t ld, const double* __restrict__ Wk, double* __restrict__ out, int64_t ldo) {
    constexpr int NB = KB < 3 ? (KB == 0 ? 3 : 4 - KB) : 1;        // staged blocks: X_KB, then L[KB+1,KB] (, L[KB+2,KB])
    const int tid = threadIdx.x;
    f64x2 v[NB][2];
#pragma unroll
    for (int bq = 0; bq < NB; ++bq) {
        const double* src = bq == 0 ? Wk + (int64_t)(32 * KB) * TS + 32 * KB : Lkk + (int64_t)(32 * (KB + bq)) * ld + 32 * KB;
        const int64_t sld = bq == 0 ? TS : ld;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * NT, r = e >> 4, c = (e & 15) * 2;
            v[bq][u] = *reinterpret_cast<const f64x2*>(src + (int64_t)r * sld + c);
        }
    }
    Trsm64Ops o;
    if constexpr (KB == 0) blk64_load(o.bl[2], Lkk + (int64_t)(32 * 3) * ld, ld);
#pragma unroll
    for (int bq = 0; bq < NB; ++bq)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * NT, r = e >> 4, c = (e & 15) * 2;
            *reinterpret_cast<f64x2*>(stage + bq * OPS_BLOCK + r * OPS_LD + c) = v[bq][u];
        }
    __syncthreads();
    blk64_from_stage(o.bx, stage);
#pragma unroll
    for (int bq = 1; bq < NB; ++bq) blk64_from_stage(o.bl[bq - 1], stage + bq * OPS_BLOCK);
    trsm64_apply<KB, STORE>(sT, o, out, ldo);
}
// slice KB of the whole 128-row tile out of the image, by ONE wave: 32 x 16-byte stores per lane, the LDS reads of FL in flight
// (FL = 8 raised the one-launch kernel's register count from 418 to 444, and the compiler then broke its main K loop into
// pieces -- 17.8 us per K block instead of 14.8 on every tile of the launch)
// WT: written through (sc1) -- the word that announces the slice then needs the wave's own drain only, no L2-wide write-back (which
// took ~6 us here, with every other CU of the XCD storing tiles)
template <int KB, bool WT, int FL = 4>
__device__ __forceinline__ void slice64_out(const double* __restrict__ sT, double* __restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63, c = (lane & 15) * 2;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7fffffff, 0x00020000);
#pragma unroll 1
    for (int it0 = 0; it0 < 32; it0 += FL) {
        f64x2 v[FL];
#pragma unroll
        for (int u = 0; u < FL; ++u) {
            const int r = (it0 + u) * 4 + (lane >> 4);
            v[u][0] = sT[r * DT64 + 32 * KB + c];
            v[u][1] = sT[r * DT64 + 32 * KB + c + 1];
        }
#pragma unroll
        for (int u = 0; u < FL; ++u) {
            const int r = (it0 + u) * 4 + (lane >> 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_64, v[u]), rs, (int)((r * ldo + 32 * KB + c) * 8), 0, WT ? 16 : 0);
        }
    }
}
// ---- the diagonal tile's sum while the tile beside it is solved (batch64_step.hip, D pieces) ------------------------------------
// Only the lower triangle of A[i,i] -= Lt Lt^T is needed: 36 of the 64 16x16 tiles.  Waves 0, 2, 3 own 12 each (wave 1 writes
// the finished slices out meanwhile): rows of tiles 0..3 and (4,0), (4,1) | (4,2) .. (4,4), row 5, (6,0) .. (6,2) | (6,3) ..
// (6,6), row 7.  Through round 5 three waves owned a 64x64 quadrant each -- 16 tiles, and the update was 3.4 of a step's 6.4 us
// at the fp64 MFMA rate (one v_mfma_f64_16x16x4 per 64 clocks and SIMD).
constexpr int DIAG_OWN = 12;
__device__ __forceinline__ int diag_group() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) == 0 ? 0 : (int)(threadIdx.x >> 6) - 1); }
__device__ __forceinline__ void diag_tile(int g, int t, int& tr, int& tc) {      // t: compile-time after unrolling, g: wave-uniform
    constexpr int R[3][DIAG_OWN] = {{0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4}, {4, 4, 4, 5, 5, 5, 5, 5, 5, 6, 6, 6}, {6, 6, 6, 6, 7, 7, 7, 7, 7, 7, 7, 7}};
    constexpr int C[3][DIAG_OWN] = {{0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1}, {2, 3, 4, 0, 1, 2, 3, 4, 5, 0, 1, 2}, {3, 4, 5, 6, 0, 1, 2, 3, 4, 5, 6, 7}};
    tr = g == 0 ? R[0][t] : (g == 1 ? R[1][t] : R[2][t]);
    tc = g == 0 ? C[0][t] : (g == 1 ? C[1][t] : C[2][t]);
}
// -(the tile at C, ld doubles) into the owners' accumulators
__device__ __forceinline__ void diag_load_neg(f64x4 (&v)[DIAG_OWN], const double* __restrict__ C, int64_t ld) {
    const int lane = threadIdx.x & 63, g = diag_group();
#pragma unroll
    for (int t = 0; t < DIAG_OWN; ++t) {
        int tr, tc;
        diag_tile(g, t, tr, tc);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[t][q] = -C[(int64_t)(16 * tr + (lane >> 4) + 4 * q) * ld + 16 * tc + (lane & 15)];
    }
}
// += Lt[:, slice KB] Lt[:, slice KB]^T from the image (call on waves 0, 2, 3)
template <int KB>
__device__ __forceinline__ void diag_syrk_slice(f64x4 (&acc)[DIAG_OWN], const double* __restrict__ sT) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, lk = lane >> 4, g = diag_group();
    const double* base = sT + l15 * DT64 + 32 * KB + lk;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        double a[DIAG_OWN], bb[DIAG_OWN];
#pragma unroll
        for (int t = 0; t < DIAG_OWN; ++t) {
            int tr, tc;
            diag_tile(g, t, tr, tc);
            a[t] = base[(16 * tr) * DT64 + 4 * st];
            bb[t] = base[(16 * tc) * DT64 + 4 * st];
        }
#pragma unroll
        for (int t = 0; t < DIAG_OWN; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], bb[t], acc[t], 0, 0, 0);
    }
}
// -acc into the LDS image, the lower triangle; zeros above the diagonal inside the four 32x32 diagonal sub-blocks (what the
// pivot phases read) -- the rest of the upper triangle is never read
__device__ __forceinline__ void diag_to_image(const f64x4 (&v)[DIAG_OWN], double* __restrict__ sT) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) sT[(32 * u + (lane >> 4) + 4 * q) * DT64 + 32 * u + 16 + (lane & 15)] = 0.0;
        return;
    }
    const int g = diag_group();
#pragma unroll
    for (int t = 0; t < DIAG_OWN; ++t) {
        int tr, tc;
        diag_tile(g, t, tr, tc);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 16 * tr + (lane >> 4) + 4 * q, c = 16 * tc + (lane & 15);
            sT[r * DT64 + c] = (c <= r) ? -v[t][q] : 0.0;
        }
    }
}

// 16 bytes out; WT: written through (two sc1 stores) -- what a hand-off word announces then needs the storing waves' drain only
template <bool WT>
__device__ __forceinline__ void store64x2(double* __restrict__ p, double a, double b) {
    if constexpr (WT) {
        __hip_atomic_store(p, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *reinterpret_cast<f64x2*>(p) = f64x2{a, b};
    }
}
#define VOLT_STAMP64(i)                                                                \
    do {                                                                               \
        if (STAMP && stamps && threadIdx.x == 0) stamps[32 * b + (i)] = __builtin_amdgcn_s_memrealtime();   \
    } while (0)
// The diagonal block of matrix b, block column k.  image_ready: the caller has already put the (lower triangle of the)
// block into the image sT (batch64_step.hip: the tile's last update lands there instead of in memory).
// sub (batch64_step.hip; LOCALPUB as in common.h): PROGRESSIVE hand-off.  The tiles below this block do not wait for the whole
// inverse W: they solve against L_kk 32 columns at a time (trsm64_step) and need, for sub-block column kb, the final blocks
// L[kb..3, kb] and X_kb = L[kb,kb]^-1.  With sub != nullptr those go to memory as soon as they exist -- the column slice
// behind its pivot phase, X_kb (into the diagonal block of W where it stays) behind the phase that inverts it -- and *sub is
// raised to kb + 1 when slice kb and X_kb are there; the separate pass that writes the off-diagonal blocks of L is gone.
template <bool STAMP, bool LOCALPUB = false>
__device__ __forceinline__ void diag64_body(double* __restrict__ A, double* __restrict__ Winv, int* __restrict__ info, int Np,
                                            int k, int b, double* __restrict__ sT, long long* stamps, bool image_ready,
                                            int* sub = nullptr) {
    // (the progressive hand-off's stores written through, and the word without a release, were measured for readers outside the
    // XCD: the drain of sc1 stores costs what the L2 write-back did -- sub-block 0 is seen 7 us after its barrier either way)
    constexpr bool WTP = false;
    VOLT_STAMP64(0);
    if (STAMP && stamps && threadIdx.x == 0) stamps[32 * b + 30] = __builtin_amdgcn_s_memtime();     // shader clocks, against [0] .. [20]'s 100 MHz
    double* colbuf = sT + TS * DT64;       // 128 doubles: the reciprocal pivots
    const int n = Np / TS, tid = threadIdx.x, wave = tid >> 6;
    double* D = A + (int64_t)b * Np * Np + (int64_t)k * TS * Np + (int64_t)k * TS;
    double* W = Winv + ((int64_t)b * n + k) * TS * TS;
    if (!image_ready) {   // lower triangle in, 8 x 16-byte loads per thread in flight (one load per iteration cost 7 us of latency)
        constexpr int PER = TS * TS / 2 / NT;                     // 32 double pairs per thread
#pragma unroll
        for (int it0 = 0; it0 < PER; it0 += 8) {
            f64x2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + (it0 + u) * NT, r = e >> 6, c = (e & 63) * 2;
                v[u] = f64x2{0.0, 0.0};
                if (c <= r) v[u] = *reinterpret_cast<const f64x2*>(D + (int64_t)r * Np + c);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + (it0 + u) * NT, r = e >> 6, c = (e & 63) * 2;
                sT[r * DT64 + c] = v[u][0];
                sT[r * DT64 + c + 1] = (c + 1 <= r) ? v[u][1] : 0.0;
            }
        }
    }
    __syncthreads();
    VOLT_STAMP64(1);
    int bad = 0;
    double* rdiag = colbuf;                // 128 reciprocal pivots
    // Sub-block column kb: pivot waves w < max(1, 3 - kb) factor (kb,kb) with the panel block (kb+1+w, kb) riding along.  Then
    // wave 3 inverts the sub-block (1.8 us) WHILE waves 0..2 do the trailing updates of the blocks to its right (6, 3, 1, 0 block
    // products), so that X_kb goes out -- and, progressive, sub is raised -- one whole pivot phase earlier than through round 5,
    // where the inverse rode along with the NEXT pivot phase: the tile below starts its solve 6 us sooner.
    for (int kb = 0; kb < 4; ++kb) {
        const int npw = kb < 3 ? 3 - kb : 1;
        if (wave < npw) pivot_phase64(sT, rdiag, kb, kb + 1 + wave <= 3 ? kb + 1 + wave : -1, wave == 0, bad);
        if (STAMP && stamps && kb == 1 && (threadIdx.x & 63) == 0) stamps[32 * b + 21 + wave] = __builtin_amdgcn_s_memrealtime();   // who the barrier waits for
        __syncthreads();
        VOLT_STAMP64(2 + 4 * kb);
        if (wave == 3) {
            // L_kk out (zeros above the diagonal) by the wave that is about to overwrite it, then X_kb in its place and out
            const int lane = tid & 63;
            for (int e = lane; e < 32 * 16; e += 64) {
                const int r = e >> 4, c = (e & 15) * 2;
                const double* s = sT + (32 * kb + r) * DT64 + 32 * kb + c;
                store64x2<WTP>(D + (int64_t)(32 * kb + r) * Np + 32 * kb + c, s[0], s[1]);
            }
            inv32_f64(sT, rdiag, kb);
            if (sub) {                                        // (the wave's own LDS writes: in order, no barrier)
                for (int e = lane; e < 32 * 16; e += 64) {
                    const int r = e >> 4, c = (e & 15) * 2;
                    const double* s = sT + (32 * kb + r) * DT64 + 32 * kb + c;
                    store64x2<WTP>(W + (int64_t)(32 * kb + r) * TS + 32 * kb + c, s[0], s[1]);
                }
            }
        } else {
            // progressive: the blocks below L_kk, which rode along and are final too
            if (sub) {
                for (int e = tid; e < (TS - 32 * kb - 32) * 16; e += NT - 64) {
                    const int r = 32 + (e >> 4), c = (e & 15) * 2;
                    const double* s = sT + (32 * kb + r) * DT64 + 32 * kb + c;
                    store64x2<WTP>(D + (int64_t)(32 * kb + r) * Np + 32 * kb + c, s[0], s[1]);
                }
            }
            // trailing updates A[i,j] -= L[i,kb] L[j,kb]^T, kb < j <= i <= 3, dealt round-robin to waves 0..2
            int cnt = 0;
            for (int i = kb + 1; i <= 3; ++i)
                for (int j = kb + 1; j <= i; ++j) {
                    if (wave == (cnt++ % 3)) {
                        double* C = sT + (32 * i) * DT64 + 32 * j;
                        f64x4 acc[4];
                        acc64_load(acc, C);
                        mm64_nt<true>(acc, sT + (32 * i) * DT64 + 32 * kb, sT + (32 * j) * DT64 + 32 * kb);
                        acc64_store(acc, C, 1.0);
                    }
                }
        }
        VOLT_STAMP64(3 + 4 * kb);
        // (drain, barrier; release + word by wave 3, which has nothing to do in the next pivot phase)
        if (sub) batch_publish_release<LOCALPUB>(sub, kb + 1, 192);
        else __syncthreads();
        VOLT_STAMP64(5 + 4 * kb);
    }
    // off-diagonal L blocks out (the diagonal sub-blocks went out above), zeros above the diagonal: 16-byte stores, the
    // LDS reads of 8 of them in flight at a time
    if (!sub) {
        constexpr int PER = TS * TS / 2 / NT;
#pragma unroll
        for (int it0 = 0; it0 < PER; it0 += 8) {
            f64x2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + (it0 + u) * NT, r = e >> 6, c = (e & 63) * 2;
                v[u][0] = (c < r) ? sT[r * DT64 + c] : 0.0;
                v[u][1] = (c + 1 < r) ? sT[r * DT64 + c + 1] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + (it0 + u) * NT, r = e >> 6, c = (e & 63) * 2;
                if ((r >> 5) != (c >> 5)) *reinterpret_cast<f64x2*>(D + (int64_t)r * Np + c) = v[u];
            }
        }
    }
    VOLT_STAMP64(18);
    // ---- W = L^-1, blocked by 32: wave j < 3 owns block column j (the diagonal blocks of W are the X_kb in place)
    f64x4 Wr[3][4];
    if (wave < 3) {
        const int j = wave;
#pragma unroll
        for (int di = 1; di <= 3; ++di) {
            const int i = j + di;
            if (i <= 3) {                                                     // wave-uniform
                f64x4 S[4];
                acc64_zero(S);
                mm64_nn(S, sT + (32 * i) * DT64 + 32 * j, sT + (32 * j) * DT64 + 32 * j);            // L[i,j] X_j
#pragma unroll
                for (int dm = 1; dm < di; ++dm)
                    mm64_lds_reg(S, sT + (32 * i) * DT64 + 32 * (j + dm), Wr[dm - 1]);              // L[i,m] W[m,j]
                f64x4 R[4];
                acc64_zero(R);
                mm64_lds_reg(R, sT + (32 * i) * DT64 + 32 * i, S);                                   // X_i S
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) Wr[di - 1][t][q] = -R[t][q];
            }
        }
    }
    __syncthreads();                                                          // every L block has been consumed
    VOLT_STAMP64(19);
    if (wave < 3) {
        const int j = wave;
#pragma unroll
        for (int di = 1; di <= 3; ++di) {
            const int i = j + di;
            if (i <= 3) acc64_store(Wr[di - 1], sT + (32 * i) * DT64 + 32 * j, 1.0);
        }
    }
    __syncthreads();
    {
        constexpr int PER = TS * TS / 2 / NT;
#pragma unroll
        for (int it0 = 0; it0 < PER; it0 += 8) {
            f64x2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + (it0 + u) * NT, r = e >> 6, c = (e & 63) * 2;
                v[u][0] = (c <= r) ? sT[r * DT64 + c] : 0.0;
                v[u][1] = (c + 1 <= r) ? sT[r * DT64 + c + 1] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + (it0 + u) * NT, r = e >> 6, c = (e & 63) * 2;
                *reinterpret_cast<f64x2*>(W + r * TS + c) = v[u];
            }
        }
    }
    if (tid == 0 && bad) atomicCAS(info + b, 0, k * TS + bad);
    VOLT_STAMP64(20);
    if (STAMP && stamps && threadIdx.x == 0) stamps[32 * b + 31] = __builtin_amdgcn_s_memtime();
}

// Y[i,i] = W_i^T (upper triangular), through a 64 x 129-double LDS image at a time (66 KB of smem)
__device__ __forceinline__ void trtri64_diag_body(const double* __restrict__ W, double* __restrict__ Yd, int Np, float* smem) {
    double* sW = reinterpret_cast<double*>(smem);
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * TS; e += NT) {
            const int r = e >> 7, c = e & 127;                 // W row 64 half + r, column c
            sW[r * (TS + 1) + c] = W[(int64_t)(64 * half + r) * TS + c];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < TS * 64; e += NT) {
            const int c = e >> 6, r = e & 63;                  // Y row c, column 64 half + r
            Yd[(int64_t)c * Np + 64 * half + r] = (64 * half + r >= c) ? sW[r * (TS + 1) + c] : 0.0;
        }
    }
}

}  // namespace volt
