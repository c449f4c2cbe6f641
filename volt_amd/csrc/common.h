// Shared device code for libvolt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VOLT_ABI_VERSION 5

namespace volt {

constexpr int TS = 128;        // tile edge of every blocked algorithm here
constexpr int BK = 32;         // K-chunk staged through LDS per pipeline step
constexpr int SLD = BK + 4;    // LDS row stride (floats): 144 B rows keep ds_read_b128 conflict-free
constexpr int NT = 256;        // threads per workgroup (4 wave64)
constexpr int STAGE_FLOATS = 2 * TS * SLD;             // one A tile + one B tile
constexpr int GEMM_LDS_BYTES = 2 * STAGE_FLOATS * 4;   // double buffered: 73,728 B -> 2 WG / CU

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VOLT_LAUNCH_CHECK()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return (int)e__;               \
    } while (0)

// Row of a 32x32 MFMA accumulator element: reg q of lane l sits at (row, col) = (accrow(q,l), l&31).
__device__ __forceinline__ int accrow(int q, int lane) { return (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5); }

// XCD-aware decode of a 1-D grid into (tile, batch).  The dispatcher places workgroup w on XCD
// w % 8 (observed, used for speed only): give every XCD whole matrices so the operand a panel's
// tiles share (the k-th block row, or a diagonal inverse) is fetched into one L2, not eight.
__device__ __forceinline__ void decode_tile_batch(int w, int ntiles, int nbatch, int& tile, int& batch) {
    if ((nbatch & 7) == 0) {
        const int xcd = w & 7, slot = w >> 3;
        batch = (slot / ntiles) * 8 + xcd;
        tile = slot % ntiles;
    } else {
        batch = w / ntiles;
        tile = w % ntiles;
    }
}

__device__ __forceinline__ void decode_tile_batch(int ntiles, int nbatch, int& tile, int& batch) {
    decode_tile_batch((int)blockIdx.x, ntiles, nbatch, tile, batch);
}

// ---------------------------------------------------------------------------------------------
// 128x128 (x K) "NT" GEMM core on fp32 MFMA:  acc[r][c] += sum_k Arows[r][k] * Brows[c][k]
// Both operands are row-major with K contiguous (rows of L / Y / W), which is the only product
// shape the blocked Cholesky, the TRSM-by-inverse and the triangular inverse need.
//
// WL = 0: waves in a 2x2 grid, each owning 64x64 = acc[tm*2+tn] (32x32 sub-tiles).
// WL = 1: waves side by side, wave w owning all 128 rows x 32 columns [32w,32w+32) = acc[tm].
//
// Pipeline: register-staged prefetch (8 x global_load_dwordx4 per thread and chunk), double-buffered
// LDS, one barrier per chunk.  LDS rows are padded to 36 floats so the ds_read_b128 fragment reads (lane = row) hit 16
// distinct 16-byte slots per lane group.  Each b128 read feeds four v_mfma_f32_32x32x2_f32: lane
// halves take k = 4*(lane>>5)+m, consistently for A and B, so the sum over k is complete.
// ---------------------------------------------------------------------------------------------
struct StageRegs {
    f32x4 a[4], b[4];
};

__device__ __forceinline__ void stage_load(StageRegs& s, const float* __restrict__ A, int64_t lda,
                                           const float* __restrict__ B, int64_t ldb, int k0) {
    const int t = threadIdx.x;
    const int row = t >> 3, cq = (t & 7) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        s.a[p] = *reinterpret_cast<const f32x4*>(A + (int64_t)(row + 32 * p) * lda + k0 + cq);
        s.b[p] = *reinterpret_cast<const f32x4*>(B + (int64_t)(row + 32 * p) * ldb + k0 + cq);
    }
}

__device__ __forceinline__ void stage_store(const StageRegs& s, float* __restrict__ buf) {
    const int t = threadIdx.x;
    const int row = t >> 3, cq = (t & 7) * 4;
    float* sA = buf;
    float* sB = buf + TS * SLD;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        *reinterpret_cast<f32x4*>(sA + (row + 32 * p) * SLD + cq) = s.a[p];
        *reinterpret_cast<f32x4*>(sB + (row + 32 * p) * SLD + cq) = s.b[p];
    }
}

// Buffer-addressed staging: one 128-bit resource per operand (wave-uniform), four per-lane byte
// offsets computed once, the K offset in an SGPR -- no 64-bit VALU address arithmetic per chunk.
struct StageAddr {
    __amdgpu_buffer_rsrc_t ra, rb;
    int va[4], vb[4];
};

__device__ __forceinline__ StageAddr stage_addr(const float* A, int64_t lda, const float* B, int64_t ldb) {
    StageAddr sa;
    sa.ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
    sa.rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000);
    const int t = threadIdx.x;
    const int row = t >> 3, cq = (t & 7) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        sa.va[p] = (int)(((int64_t)(row + 32 * p) * lda + cq) * 4);
        sa.vb[p] = (int)(((int64_t)(row + 32 * p) * ldb + cq) * 4);
    }
    return sa;
}

__device__ __forceinline__ void stage_load_buf(StageRegs& s, const StageAddr& sa, int k0) {
    const int so = __builtin_amdgcn_readfirstlane(k0 * 4);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        s.a[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sa.ra, sa.va[p], so, 0));
        s.b[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sa.rb, sa.vb[p], so, 0));
    }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
}

// ---- the K loop: fragments double-buffered in registers ---------------------------------------------------------
// A chunk is 4 K steps of 8; step kk+1's fragments are requested BEFORE step kk's 16 MFMAs, into a second register
// set, and the chunk's barrier sits before its last step, so the first fragments of the next chunk are in flight
// during that step's MFMAs as well.  (Round 1 re-used ONE fragment set: the ds_reads of a step could only be issued
// behind the 15th MFMA of the previous one, their latency was exposed once per step, and a lone wave per SIMD reached
// 81 % MFMA duty -- ISA inspection; 132 -> 137 TF/s at k = 16, 120 -> 136 at k = 28 for the panel update alone.)
template <int WL> struct Frag;
template <> struct Frag<0> { f32x4 a0, a1, b0, b1; };
template <> struct Frag<1> { f32x4 a[4], b0; };

template <int WL>
__device__ __forceinline__ void frag_load(Frag<WL>& f, const float* __restrict__ buf, int kk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const float* sA = buf;
    const float* sB = buf + TS * SLD;
    const int ko = kk * 8 + 4 * lh;
    if constexpr (WL == 0) {
        const int wr = wave >> 1, wc = wave & 1;
        f.a0 = *reinterpret_cast<const f32x4*>(sA + (wr * 64 + l31) * SLD + ko);
        f.b0 = *reinterpret_cast<const f32x4*>(sB + (wc * 64 + l31) * SLD + ko);
        f.b1 = *reinterpret_cast<const f32x4*>(sB + (wc * 64 + 32 + l31) * SLD + ko);
        f.a1 = *reinterpret_cast<const f32x4*>(sA + (wr * 64 + 32 + l31) * SLD + ko);
    } else {
        f.b0 = *reinterpret_cast<const f32x4*>(sB + (wave * 32 + l31) * SLD + ko);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) f.a[tm] = *reinterpret_cast<const f32x4*>(sA + (tm * 32 + l31) * SLD + ko);
    }
}

// the 4 MFMAs of K sub-step m (of 4) of a fragment set
template <int WL>
__device__ __forceinline__ void frag_mma_m(const Frag<WL>& f, int m, f32x16 (&acc)[4]) {
    if constexpr (WL == 0) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[m], f.b0[m], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[m], f.b1[m], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[m], f.b0[m], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[m], f.b1[m], acc[3], 0, 0, 0);
    } else {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[tm][m], f.b0[m], acc[tm], 0, 0, 0);
    }
}
template <int WL>
__device__ __forceinline__ void frag_mma(const Frag<WL>& f, f32x16 (&acc)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) frag_mma_m<WL>(f, m, acc);
}

// ---- two-level summation (round 4) ------------------------------------------------------------------------------
// A left-looking tile sums K = 128 k products per element.  As ONE fp32 chain that starts at -A (what rounds 1-3 did)
// every one of those K roundings happens at the magnitude of A, and the factor came out 5 - 30 x further from the fp64
// factor than the vendor's blocked right-looking potrf (scripts/acc_diag.py: the ratio grows with the block column),
// whose trailing update rounds at that magnitude once per block step.  The fp32 MFMA itself is an exact chain of RNE
// FMAs (scripts/ubench/mfma_round.hip), so the remedy is the order of summation: the products of SEG_CHUNKS chunks (128
// of K) are summed from ZERO in a second accumulator set -- the first MFMAs of a segment take the constant 0 as C -- and
// the segment is then added to the running sum (64 v_add per thread and segment).
#ifndef VOLT_SEG_CHUNKS
#define VOLT_SEG_CHUNKS 4
#endif
#ifndef VOLT_SEG_TRI
#define VOLT_SEG_TRI 1
#endif
#ifndef VOLT_SEG_SQ
#define VOLT_SEG_SQ 1
#endif
constexpr int SEG_CHUNKS = VOLT_SEG_CHUNKS;
__device__ __forceinline__ f32x16 zero16c() {
    f32x16 z;
#pragma unroll
    for (int q = 0; q < 16; ++q) z[q] = 0.f;
    return z;
}
// K sub-step 0 of a fragment set into accumulators that start from zero
template <int WL>
__device__ __forceinline__ void frag_mma_m0_zero(const Frag<WL>& f, f32x16 (&acc)[4]) {
    const f32x16 z = zero16c();
    if constexpr (WL == 0) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[0], f.b0[0], z, 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[0], f.b1[0], z, 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[0], f.b0[0], z, 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[0], f.b1[0], z, 0, 0, 0);
    } else {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[tm][0], f.b0[0], z, 0, 0, 0);
    }
}
template <int WL, bool FIRST>
__device__ __forceinline__ void frag_mma_seg(const Frag<WL>& f, f32x16 (&acc)[4]) {
    if constexpr (FIRST) {
        frag_mma_m0_zero<WL>(f, acc);
#pragma unroll
        for (int m = 1; m < 4; ++m) frag_mma_m<WL>(f, m, acc);
    } else {
        frag_mma<WL>(f, acc);
    }
}
__device__ __forceinline__ void seg_flush(f32x16 (&T)[4], const f32x16 (&P)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) T[i] += P[i];
}

#define VOLT_SB() __builtin_amdgcn_sched_barrier(0)

// piece p (of 4) of a staging action: one row group of A and of B
__device__ __forceinline__ void stage_store_piece(const StageRegs& s, float* __restrict__ buf, int p) {
    const int t = threadIdx.x;
    const int row = t >> 3, cq = (t & 7) * 4;
    *reinterpret_cast<f32x4*>(buf + (row + 32 * p) * SLD + cq) = s.a[p];
    *reinterpret_cast<f32x4*>(buf + TS * SLD + (row + 32 * p) * SLD + cq) = s.b[p];
}
__device__ __forceinline__ void stage_load_piece(StageRegs& s, const StageAddr& sa, int so, int p) {
    s.a[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sa.ra, sa.va[p], so, 0));
    s.b[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sa.rb, sa.vb[p], so, 0));
}

// One chunk of the v2 pipeline.  On entry F0 holds K step 0 of `cur`; on exit F0 holds step 0 of `nxt` (if `more`).
// The order below is PINNED with sched_barriers (left alone, the machine scheduler sinks every ds_read to just
// before its first use): each step's fragment reads are issued a full step (16 MFMAs = 1024 cycles) ahead, and the
// staging traffic -- LDS write of chunk c+1 behind step 1 (`st`, if do_st), global loads of chunk c+3 behind step 2
// (`ld`, if do_ld) -- goes out two instructions at a time between groups of four MFMAs.
template <int WL, bool STEADY, bool FIRST = false>
__device__ __forceinline__ void chunk_run(const float* cur, float* nxt, bool more_, Frag<WL>& F0, Frag<WL>& F1,
                                         f32x16 (&acc)[4], StageRegs& s, bool do_st_, bool do_ld_, const StageAddr& sa,
                                         int k_ld) {
    // STEADY: everything is on (compile-time), so the loop body is one basic block
    const bool more = STEADY || more_, do_st = STEADY || do_st_, do_ld = STEADY || do_ld_;
    frag_load<WL>(F1, cur, 1);
    VOLT_SB();
    frag_mma_seg<WL, FIRST>(F0, acc);
    VOLT_SB();
    frag_load<WL>(F0, cur, 2);
    VOLT_SB();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        frag_mma_m<WL>(F1, m, acc);
        if (do_st) stage_store_piece(s, nxt, m);
        VOLT_SB();
    }
    frag_load<WL>(F1, cur, 3);
    VOLT_SB();
    const int so = __builtin_amdgcn_readfirstlane(k_ld * 4);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        frag_mma_m<WL>(F0, m, acc);
        if (do_ld) stage_load_piece(s, sa, so, m);
        VOLT_SB();
    }
    __syncthreads();
    if (more) frag_load<WL>(F0, nxt, 0);
    VOLT_SB();
    frag_mma<WL>(F1, acc);
    VOLT_SB();
}

template <int WL>
__device__ __forceinline__ void gemm_nt_128(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                               int64_t ldb, int nchunks, f32x16 (&acc)[4], float* smem) {
    if (nchunks <= 0) return;
    StageRegs s0, s1;
    const StageAddr sa = stage_addr(A, lda, B, ldb);
    stage_load_buf(s0, sa, 0);
    stage_store(s0, smem);
    if (nchunks > 1) stage_load_buf(s0, sa, BK);
    if (nchunks > 2) stage_load_buf(s1, sa, 2 * BK);
    __syncthreads();
    Frag<WL> F0, F1;
    frag_load<WL>(F0, smem, 0);
    float* b0 = smem;
    float* b1 = smem + STAGE_FLOATS;
    f32x16 P[4];                                // the segment's products, summed from zero (two-level summation, above)
    int c = 0;
#if VOLT_SEG_SQ
    for (; c + SEG_CHUNKS + 4 <= nchunks; c += SEG_CHUNKS) {   // steady state: every load / store / next-chunk read exists
        chunk_run<WL, true, true>(b0, b1, true, F0, F1, P, s0, true, true, sa, (c + 3) * BK);
        chunk_run<WL, true>(b1, b0, true, F0, F1, P, s1, true, true, sa, (c + 4) * BK);
#pragma unroll
        for (int u = 2; u < SEG_CHUNKS; u += 2) {
            chunk_run<WL, true>(b0, b1, true, F0, F1, P, s0, true, true, sa, (c + u + 3) * BK);
            chunk_run<WL, true>(b1, b0, true, F0, F1, P, s1, true, true, sa, (c + u + 4) * BK);
        }
        seg_flush(acc, P);
    }
    for (; c < nchunks; c += SEG_CHUNKS) {      // the last segment (and any chunk count that is not a multiple of it)
        // chunk c (b0): write chunk c+1 (s0) into b1, request chunk c+3 into s0
        chunk_run<WL, false, true>(b0, b1, c + 1 < nchunks, F0, F1, P, s0, c + 1 < nchunks, c + 3 < nchunks, sa, (c + 3) * BK);
        // chunk c+1 (b1): write chunk c+2 (s1) into b0, request chunk c+4 into s1
        if (c + 1 < nchunks)
            chunk_run<WL, false>(b1, b0, c + 2 < nchunks, F0, F1, P, s1, c + 2 < nchunks, c + 4 < nchunks, sa, (c + 4) * BK);
        for (int u = 2; u < SEG_CHUNKS && c + u < nchunks; u += 2) {
            chunk_run<WL, false>(b0, b1, c + u + 1 < nchunks, F0, F1, P, s0, c + u + 1 < nchunks, c + u + 3 < nchunks, sa, (c + u + 3) * BK);
            if (c + u + 1 < nchunks)
                chunk_run<WL, false>(b1, b0, c + u + 2 < nchunks, F0, F1, P, s1, c + u + 2 < nchunks, c + u + 4 < nchunks, sa, (c + u + 4) * BK);
        }
        seg_flush(acc, P);
    }
#else
    (void)P;
    for (; c + 4 < nchunks; c += 2) {           // steady state: every load / store / next-chunk read exists
        chunk_run<WL, true>(b0, b1, true, F0, F1, acc, s0, true, true, sa, (c + 3) * BK);
        chunk_run<WL, true>(b1, b0, true, F0, F1, acc, s1, true, true, sa, (c + 4) * BK);
    }
    for (; c + 1 < nchunks; c += 2) {           // the last <= 4 chunks
        chunk_run<WL, false>(b0, b1, true, F0, F1, acc, s0, true, c + 3 < nchunks, sa, (c + 3) * BK);
        chunk_run<WL, false>(b1, b0, c + 2 < nchunks, F0, F1, acc, s1, c + 2 < nchunks, c + 4 < nchunks, sa, (c + 4) * BK);
    }
    if (c < nchunks) chunk_run<WL, false>(b0, b1, false, F0, F1, acc, s0, false, false, sa, 0);
#endif
    __syncthreads();                           // smem is free for reuse on return
}

// ---- two-phase tile:  T += X Z^T  (K = 32 n1), then  O[c][r] = sum_p T[p][c] W[r][p]  --------------------------------
// The shape shared by a row tile of the triangular inverse (X = L[i,:], Z = Y[j,:], W = W_i) and by the fused
// panel-update + panel-solve tile of the factorisation (X = L[k,:], Z = L[i,:], T0 = -A[i,k]^T, W = W_k).  Waves side
// by side (WL = 1: a wave owns all 128 p for its 32 columns c), so the accumulator layout of T (lane = c, registers
// = p) IS the A-operand layout of the second product and T never leaves the register file; W streams through the
// same double-buffered LDS stage as 4 more chunks of ONE pipeline (B operand only), and its row blocks above the
// diagonal (W lower triangular) are skipped.  `flag` (optional) is the word that says W has been published by
// another workgroup of the same launch: polled by one lane before the first W load is issued, then one agent-scope
// acquire; the chunk barriers that follow cover the workgroup.
struct TriTile {
    const float* X;  int64_t ldx;      // 128 rows p, K contiguous
    const float* Z;  int64_t ldz;      // 128 rows c
    int n1;                            // K / 32 (multiple of 4, may be 0)
    const float* W;                    // [128][128] row-major, lower triangular
    const int* flag;                   // nullptr, or W's ready flag
    int want;                          // 0: ready = the flag word is non-zero;  else: ready = the word equals `want`
};

// One per-lane byte offset per operand (row t >> 3, 16-byte column t & 7); the row group p (32 rows further down) and
// the K offset of the chunk are wave-uniform and ride in the instruction's SGPR offset.
// Cache policy of the operand each tile reads ONCE (Z: its own block row): nt (aux = 2), so that it does not push the
// operand the tiles of a matrix SHARE (X: block row k) out of the XCD's L2 -- FETCH_SIZE per launch 1.95 -> 1.70 GB,
// speed unchanged (scripts/nt_exp.sh).
#ifndef VOLT_Z_AUX
#define VOLT_Z_AUX 2
#endif
struct TriSrc {
    __amdgpu_buffer_rsrc_t ra, rb, rw;
    int va, vb, vw;
    int pa, pb;                        // byte stride of a row group: 32 ldx * 4, 32 ldz * 4  (W: 32 * 128 * 4)
    int n1, nall;
};

// PH1: the chunk is known (at compile time) to be a phase-1 chunk -- the steady state; otherwise decided at run time
template <bool PH1 = false>
__device__ __forceinline__ void tri_load_piece(StageRegs& s, const TriSrc& ts, int c, int p) {
    if (PH1 || c < ts.n1) {
        const int sa = __builtin_amdgcn_readfirstlane(c * BK * 4 + p * ts.pa);
        const int sb = __builtin_amdgcn_readfirstlane(c * BK * 4 + p * ts.pb);
        s.a[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ts.ra, ts.va, sa, 0));
        s.b[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ts.rb, ts.vb, sb, VOLT_Z_AUX));
    } else {
        const int so = __builtin_amdgcn_readfirstlane((c - ts.n1) * BK * 4 + p * (32 * TS * 4));
        s.b[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ts.rw, ts.vw, so, 0));
    }
}
template <bool PH1 = false>
__device__ __forceinline__ void tri_store_piece(const StageRegs& s, float* __restrict__ buf, const TriSrc& ts, int c, int p) {
    const int t = threadIdx.x;
    const int row = t >> 3, cq = (t & 7) * 4;
    if (PH1 || c < ts.n1) *reinterpret_cast<f32x4*>(buf + (row + 32 * p) * SLD + cq) = s.a[p];
    *reinterpret_cast<f32x4*>(buf + TS * SLD + (row + 32 * p) * SLD + cq) = s.b[p];
}

// phase-1 chunk `c` living in `cur`: stores chunk c+1 (from `s`) into `nxt`, requests chunk c+3 into `s`
template <bool STEADY, bool FIRST = false>
__device__ __forceinline__ void tri_chunk_p1(const float* cur, float* nxt, int c, Frag<1>& F0, Frag<1>& F1,
                                             f32x16 (&T)[4], StageRegs& s, const TriSrc& ts) {
    const bool more = STEADY || (c + 1 < ts.n1);               // is the next chunk a phase-1 chunk (Frag<1> reads)?
    const bool do_st = STEADY || (c + 1 < ts.nall), do_ld = STEADY || (c + 3 < ts.nall);
    frag_load<1>(F1, cur, 1);
    VOLT_SB();
    frag_mma_seg<1, FIRST>(F0, T);
    VOLT_SB();
    frag_load<1>(F0, cur, 2);
    VOLT_SB();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        frag_mma_m<1>(F1, m, T);
        if (do_st) tri_store_piece<STEADY>(s, nxt, ts, c + 1, m);
        VOLT_SB();
    }
    frag_load<1>(F1, cur, 3);
    VOLT_SB();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        frag_mma_m<1>(F0, m, T);
        if (do_ld) tri_load_piece<STEADY>(s, ts, c + 3, m);
        VOLT_SB();
    }
    __syncthreads();
    if (more) frag_load<1>(F0, nxt, 0);
    VOLT_SB();
    frag_mma<1>(F1, T);
    VOLT_SB();
}

// W fragments of K step g (8 values of p) for the row blocks rb >= tp
struct FragW { f32x4 w[4]; };
__device__ __forceinline__ void fragw_load(FragW& f, const float* __restrict__ buf, int tp, int g) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const float* sB = buf + TS * SLD;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
        if (rb >= tp) f.w[rb] = *reinterpret_cast<const f32x4*>(sB + (rb * 32 + l31) * SLD + 8 * g + 4 * lh);
}
// registers 4g..4g+3 of T[tp] <-> p = 32 tp + 8g + 4 lh + (0..3)
__device__ __forceinline__ void fragw_mma_m(const FragW& f, const f32x16& Ttp, int tp, int g, int m, f32x16 (&O)[4]) {
    const float a = Ttp[4 * g + m];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
        if (rb >= tp) O[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, f.w[rb][m], O[rb], 0, 0, 0);
}

// phase-2 chunk tp (compile-time: the triangular skip makes the MFMA count depend on it) living in `cur`
template <int TP>
__device__ __forceinline__ void tri_chunk_p2(const float* cur, float* nxt, f32x16 (&T)[4], f32x16 (&O)[4],
                                             StageRegs& s, const TriSrc& ts) {
    const int c = ts.n1 + TP;
    FragW G0, G1;
    fragw_load(G0, cur, TP, 0);
    fragw_load(G1, cur, TP, 1);
    VOLT_SB();
#pragma unroll
    for (int m = 0; m < 4; ++m) fragw_mma_m(G0, T[TP], TP, 0, m, O);
    VOLT_SB();
    fragw_load(G0, cur, TP, 2);
    VOLT_SB();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        fragw_mma_m(G1, T[TP], TP, 1, m, O);
        if (TP < 3) tri_store_piece(s, nxt, ts, c + 1, m);
        VOLT_SB();
    }
    fragw_load(G1, cur, TP, 3);
    VOLT_SB();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        fragw_mma_m(G0, T[TP], TP, 2, m, O);
        if (TP == 0) tri_load_piece(s, ts, c + 3, m);           // only chunk n1+3 is still to be requested
        VOLT_SB();
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m) fragw_mma_m(G1, T[TP], TP, 3, m, O);
    VOLT_SB();
}

// Hand-off waits are bounded by WALL CLOCK (s_memrealtime: a constant 100 MHz counter), not by an iteration count: a
// preempted or profiled run spins more often, not longer.  3 s -- only a bug or a wedged device gets there.
constexpr unsigned long long WAIT_LIMIT_TICKS = 300000000ull;
// `want` = 0: wait for a non-zero word;  else wait for the word to EQUAL `want` (flags that carry the number of the step
// they belong to and are never cleared: small_step_kernel in chol.hip).
__device__ __forceinline__ bool flag_is_set(const int* flag, int want) {
    const int v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return want ? v == want : v != 0;
}
__device__ __forceinline__ bool wait_flag(const int* flag, int want, int sleep) {
    if (flag_is_set(flag, want)) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned spins = 0;
    while (!flag_is_set(flag, want)) {
        if (sleep == 2) __builtin_amdgcn_s_sleep(2);
        else __builtin_amdgcn_s_sleep(4);
        if ((++spins & 1023u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS) return false;
    }
    return true;
}
__device__ __forceinline__ bool wait_nonzero(const int* flag, int sleep) { return wait_flag(flag, 0, sleep); }
__device__ __forceinline__ bool flag_wait_one_lane(const int* flag, int want = 0) {
    bool ok = true;
    if (threadIdx.x == 0) {
        ok = wait_flag(flag, want, 4);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    return ok;       // meaningful in thread 0 only
}
__device__ __forceinline__ void flag_publish(int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave drains
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // restate the wait the compiler may drop
        __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// T holds T0 on entry; O (zeroed here) holds the product on exit.  Returns false (thread 0) if the flag timed out.
__device__ __forceinline__ bool tri_tile_run(const TriTile& t, f32x16 (&T)[4], f32x16 (&O)[4], float* smem) {
    const int tid = threadIdx.x;
    const int srow = tid >> 3, scq = (tid & 7) * 4;
    TriSrc ts;
    ts.ra = __builtin_amdgcn_make_buffer_rsrc((void*)t.X, 0, 0x7fffffff, 0x00020000);
    ts.rb = __builtin_amdgcn_make_buffer_rsrc((void*)t.Z, 0, 0x7fffffff, 0x00020000);
    ts.rw = __builtin_amdgcn_make_buffer_rsrc((void*)t.W, 0, TS * TS * 4, 0x00020000);
    ts.va = (int)(((int64_t)srow * t.ldx + scq) * 4);
    ts.vb = (int)(((int64_t)srow * t.ldz + scq) * 4);
    ts.vw = (srow * TS + scq) * 4;
    ts.pa = (int)(32 * t.ldx * 4);
    ts.pb = (int)(32 * t.ldz * 4);
    ts.n1 = t.n1;
    ts.nall = t.n1 + 4;
    bool ok = true;
    if (t.flag && t.n1 == 0) {                       // no phase 1 to hide behind: W is the first thing needed
        ok = flag_wait_one_lane(t.flag, t.want);
        __syncthreads();
    }
    StageRegs s0, s1;
    float* b0 = smem;
    float* b1 = smem + STAGE_FLOATS;
#pragma unroll
    for (int p = 0; p < 4; ++p) tri_load_piece(s0, ts, 0, p);
#pragma unroll
    for (int p = 0; p < 4; ++p) tri_store_piece(s0, b0, ts, 0, p);
#pragma unroll
    for (int p = 0; p < 4; ++p) tri_load_piece(s0, ts, 1, p);     // odd chunks travel in s0, even ones in s1
#pragma unroll
    for (int p = 0; p < 4; ++p) tri_load_piece(s1, ts, 2, p);
    __syncthreads();
    int c = 0;
    if (t.n1 > 0) {
        // two-level summation (see SEG_CHUNKS): O, idle until phase 2, takes the products of one segment (4 chunks = 128
        // of K) from zero; the segment is then added to T.  n1 is a multiple of 4.
        Frag<1> F0, F1;
        frag_load<1>(F0, b0, 0);
#if VOLT_SEG_TRI
        for (; c + SEG_CHUNKS + 4 <= t.n1; c += SEG_CHUNKS) {     // steady state: phase-1 chunks whose prefetches are phase-1 too
            tri_chunk_p1<true, true>(b0, b1, c, F0, F1, O, s0, ts);
            tri_chunk_p1<true>(b1, b0, c + 1, F0, F1, O, s1, ts);
#pragma unroll
            for (int u = 2; u < SEG_CHUNKS; u += 2) {
                tri_chunk_p1<true>(b0, b1, c + u, F0, F1, O, s0, ts);
                tri_chunk_p1<true>(b1, b0, c + u + 1, F0, F1, O, s1, ts);
            }
            seg_flush(T, O);
        }
        for (; c < t.n1; c += 4) {                   // last chunks of phase 1: the W chunks come into view
            if (t.flag && c == t.n1 - 4) ok = flag_wait_one_lane(t.flag, t.want);   // barriers below order the acquire
            tri_chunk_p1<false, true>(b0, b1, c, F0, F1, O, s0, ts);
            tri_chunk_p1<false>(b1, b0, c + 1, F0, F1, O, s1, ts);
            tri_chunk_p1<false>(b0, b1, c + 2, F0, F1, O, s0, ts);
            tri_chunk_p1<false>(b1, b0, c + 3, F0, F1, O, s1, ts);
            seg_flush(T, O);
        }
#else
        for (; c + 4 < t.n1; c += 2) {               // steady state: phase-1 chunks whose prefetches are phase-1 too
            tri_chunk_p1<true>(b0, b1, c, F0, F1, T, s0, ts);
            tri_chunk_p1<true>(b1, b0, c + 1, F0, F1, T, s1, ts);
        }
        for (; c < t.n1; c += 2) {                   // last 4 chunks of phase 1: the W chunks come into view
            if (t.flag && c == t.n1 - 4) ok = flag_wait_one_lane(t.flag, t.want);   // barriers below order the acquire
            tri_chunk_p1<false>(b0, b1, c, F0, F1, T, s0, ts);
            tri_chunk_p1<false>(b1, b0, c + 1, F0, F1, T, s1, ts);
        }
#endif
    }
    zero_acc(O);
    // phase 2: chunks n1 .. n1+3 (n1 is even: chunk n1 sits in buffer 0)
    tri_chunk_p2<0>(b0, b1, T, O, s0, ts);
    tri_chunk_p2<1>(b1, b0, T, O, s1, ts);
    tri_chunk_p2<2>(b0, b1, T, O, s0, ts);
    tri_chunk_p2<3>(b1, b0, T, O, s1, ts);
    __syncthreads();                                 // smem is free for reuse on return
    return ok;
}

// The first phase alone (T += X Z^T over n1 chunks, n1 a positive multiple of 4), nothing of W touched: for a tile whose
// W is not there yet and whose owner has better things to do than wait inside the pipeline (small_step_kernel's spine);
// tri_tile_run with n1 = 0 then finishes it.  Same chunk code, same order of MFMAs as the fused pipeline.
__device__ __forceinline__ void tri_phase1_only(const TriTile& t, f32x16 (&T)[4], float* smem) {
    const int tid = threadIdx.x;
    const int srow = tid >> 3, scq = (tid & 7) * 4;
    TriSrc ts;
    ts.ra = __builtin_amdgcn_make_buffer_rsrc((void*)t.X, 0, 0x7fffffff, 0x00020000);
    ts.rb = __builtin_amdgcn_make_buffer_rsrc((void*)t.Z, 0, 0x7fffffff, 0x00020000);
    ts.rw = ts.ra;                                   // never used: no chunk index reaches n1
    ts.va = (int)(((int64_t)srow * t.ldx + scq) * 4);
    ts.vb = (int)(((int64_t)srow * t.ldz + scq) * 4);
    ts.vw = 0;
    ts.pa = (int)(32 * t.ldx * 4);
    ts.pb = (int)(32 * t.ldz * 4);
    ts.n1 = t.n1;
    ts.nall = t.n1;
    StageRegs s0, s1;
    float* b0 = smem;
    float* b1 = smem + STAGE_FLOATS;
#pragma unroll
    for (int p = 0; p < 4; ++p) tri_load_piece(s0, ts, 0, p);
#pragma unroll
    for (int p = 0; p < 4; ++p) tri_store_piece(s0, b0, ts, 0, p);
#pragma unroll
    for (int p = 0; p < 4; ++p) tri_load_piece(s0, ts, 1, p);
#pragma unroll
    for (int p = 0; p < 4; ++p) tri_load_piece(s1, ts, 2, p);
    __syncthreads();
    Frag<1> F0, F1;
    frag_load<1>(F0, b0, 0);
    int c = 0;
#if VOLT_SEG_TRI
    f32x16 P[4];                                     // two-level summation (see SEG_CHUNKS)
    for (; c + SEG_CHUNKS + 4 <= t.n1; c += SEG_CHUNKS) {
        tri_chunk_p1<true, true>(b0, b1, c, F0, F1, P, s0, ts);
        tri_chunk_p1<true>(b1, b0, c + 1, F0, F1, P, s1, ts);
#pragma unroll
        for (int u = 2; u < SEG_CHUNKS; u += 2) {
            tri_chunk_p1<true>(b0, b1, c + u, F0, F1, P, s0, ts);
            tri_chunk_p1<true>(b1, b0, c + u + 1, F0, F1, P, s1, ts);
        }
        seg_flush(T, P);
    }
    for (; c < t.n1; c += 4) {
        tri_chunk_p1<false, true>(b0, b1, c, F0, F1, P, s0, ts);
        tri_chunk_p1<false>(b1, b0, c + 1, F0, F1, P, s1, ts);
        tri_chunk_p1<false>(b0, b1, c + 2, F0, F1, P, s0, ts);
        tri_chunk_p1<false>(b1, b0, c + 3, F0, F1, P, s1, ts);
        seg_flush(T, P);
    }
#else
    for (; c + 4 < t.n1; c += 2) {
        tri_chunk_p1<true>(b0, b1, c, F0, F1, T, s0, ts);
        tri_chunk_p1<true>(b1, b0, c + 1, F0, F1, T, s1, ts);
    }
    for (; c < t.n1; c += 2) {
        tri_chunk_p1<false>(b0, b1, c, F0, F1, T, s0, ts);
        tri_chunk_p1<false>(b1, b0, c + 1, F0, F1, T, s1, ts);
    }
#endif
    __syncthreads();                                 // smem is free for reuse on return
}

// sum over the 64 lanes (DPP inside 16-lane rows, then four readlanes), result in every lane
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_f(float x) {
    x = dpp_add<0xB1>(x);     // quad_perm [1,0,3,2]
    x = dpp_add<0x4E>(x);     // quad_perm [2,3,0,1]
    x = dpp_add<0x141>(x);    // row_half_mirror
    x = dpp_add<0x140>(x);    // row_mirror: every lane of a 16-lane row now holds the row sum
    const int xi = __float_as_int(x);       // readlane is an integer builtin: bit-cast, never convert
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(xi, 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(xi, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(xi, 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(xi, 48));
    return (r0 + r1) + (r2 + r3);
}

}  // namespace volt
