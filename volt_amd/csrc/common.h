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
// FMAs (scripts/ubench/mfma_round.hip), so the remedy is the order of summation: the products of 4 chunks (128 of K)
// are summed from ZERO in a second accumulator set -- the first MFMAs of a segment take the constant 0 as C -- and the
// segment is then added to the running sum (64 v_add per thread and segment, in the shadow of MFMAs: below).
constexpr int SEG_CHUNKS = 4;                  // TS / BK: every K range handed to the pipelines below is a multiple of it
__device__ __forceinline__ f32x16 zero16c() {
    f32x16 z;
#pragma unroll
    for (int q = 0; q < 16; ++q) z[q] = 0.f;
    return z;
}
#define VOLT_SB() __builtin_amdgcn_sched_barrier(0)
// MFMA (m, t) of a fragment set: K sub-step m, accumulator t
template <int WL>
__device__ __forceinline__ f32x16 frag_mma_one(const Frag<WL>& f, int m, int t, const f32x16& c) {
    if constexpr (WL == 0) {
        const float a = (t & 2) ? f.a1[m] : f.a0[m], b = (t & 1) ? f.b1[m] : f.b0[m];
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[t][m], f.b0[m], c, 0, 0, 0);
    }
}
// The flush of a segment rides in the shadow of MFMAs (the adds of a quarter issue while the NEXT MFMA occupies the pipe,
// which accepted it only when the quarter's own last MFMA was done): three quarters behind the last three MFMAs of the
// segment, the fourth behind the first MFMA of the next one -- which is why P[3] starts as zero and is added once more
// after the last segment.  (As one block of 32 v_pk_add_f32 between two segments the flush cost 0.9 % of the step.)
// First 16 MFMAs of a segment: the products start from the constant 0
template <int WL>
__device__ __forceinline__ void seg_first_group(const Frag<WL>& f, f32x16 (&P)[4], f32x16 (&T)[4]) {
    const f32x16 z = zero16c();
    P[0] = frag_mma_one<WL>(f, 0, 0, z);
    VOLT_SB();
    T[3] += P[3];                                   // the previous segment's last quarter (zero before the first segment)
    VOLT_SB();
    P[1] = frag_mma_one<WL>(f, 0, 1, z);
    P[2] = frag_mma_one<WL>(f, 0, 2, z);
    P[3] = frag_mma_one<WL>(f, 0, 3, z);
#pragma unroll
    for (int m = 1; m < 4; ++m) frag_mma_m<WL>(f, m, P);
}
// Last 16 MFMAs of a segment
template <int WL>
__device__ __forceinline__ void seg_last_group(const Frag<WL>& f, f32x16 (&P)[4], f32x16 (&T)[4]) {
#pragma unroll
    for (int m = 0; m < 3; ++m) frag_mma_m<WL>(f, m, P);
    P[0] = frag_mma_one<WL>(f, 3, 0, P[0]);
    P[1] = frag_mma_one<WL>(f, 3, 1, P[1]);
    VOLT_SB();
    T[0] += P[0];
    VOLT_SB();
    P[2] = frag_mma_one<WL>(f, 3, 2, P[2]);
    VOLT_SB();
    T[1] += P[1];
    VOLT_SB();
    P[3] = frag_mma_one<WL>(f, 3, 3, P[3]);
    VOLT_SB();
    T[2] += P[2];
    VOLT_SB();
}
enum SegRole { SEG_MID = 0, SEG_FIRST = 1, SEG_LAST = 2 };


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
template <int WL, bool MORE, bool ST, bool LD, int ROLE = SEG_MID>
__device__ __forceinline__ void chunk_run(const float* cur, float* nxt, Frag<WL>& F0, Frag<WL>& F1, f32x16 (&acc)[4],
                                         f32x16 (&run)[4], StageRegs& s, const StageAddr& sa, int k_ld) {
    // MORE: a chunk follows (its first fragments are read behind the barrier);  ST: chunk c+1 goes from `s` into `nxt`;
    // LD: chunk c+3 is requested into `s`.  All compile-time -- a K range is a whole number of 4-chunk segments, so every
    // chunk's position relative to the end is known: the loop bodies are single basic blocks.  acc = the segment's
    // products, run = the running sum they are flushed into (ROLE: first / last chunk of a segment, see seg_first_group)
    frag_load<WL>(F1, cur, 1);
    VOLT_SB();
    if constexpr (ROLE == SEG_FIRST) seg_first_group<WL>(F0, acc, run);
    else frag_mma<WL>(F0, acc);
    VOLT_SB();
    frag_load<WL>(F0, cur, 2);
    VOLT_SB();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        frag_mma_m<WL>(F1, m, acc);
        if constexpr (ST) stage_store_piece(s, nxt, m);
        VOLT_SB();
    }
    frag_load<WL>(F1, cur, 3);
    VOLT_SB();
    const int so = __builtin_amdgcn_readfirstlane(k_ld * 4);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        frag_mma_m<WL>(F0, m, acc);
        if constexpr (LD) stage_load_piece(s, sa, so, m);
        VOLT_SB();
    }
    __syncthreads();
    if constexpr (MORE) frag_load<WL>(F0, nxt, 0);
    VOLT_SB();
    if constexpr (ROLE == SEG_LAST) seg_last_group<WL>(F1, acc, run);
    else frag_mma<WL>(F1, acc);
    VOLT_SB();
}

// ---- chasing tiles (batch_step.hip: the whole batched step in ONE launch) ------------------------------------------
// A left-looking tile reads K blocks 0 .. k-1 of two block rows that other workgroups of the SAME launch are still
// producing.  Instead of waiting for all of them before it starts, a chasing tile follows two progress words ("blocks
// [0, *p - base) of this operand are complete") and asks only at the top of a 128-wide K segment, for the segment whose
// loads it is about to issue -- so the long products run as far ahead as their inputs allow and what is left on the
// critical path behind a finished block column is ONE K block, as in a right-looking sweep, with the left-looking
// traffic.  Every wave polls for itself (the chunk barriers keep the waves of a workgroup within a chunk of each other,
// and each has seen for itself that what it loads is there); a tile that finds everything complete on entry -- the
// common case in a large batch -- never polls again.
#ifndef VOLT_WAIT_TICKS
#define VOLT_WAIT_TICKS 300000000ull
#endif
constexpr unsigned long long CHASE_LIMIT_TICKS = VOLT_WAIT_TICKS;   // 3 s of the 100 MHz s_memrealtime counter (WAIT_LIMIT_TICKS below)
// A poll of a hand-off word: an sc1 load -- past this CU's L1, served by the XCD's L2 (or by memory when the line was
// dropped there by a written-through store).
// LOCALP, the template switch of the waits below: the writer of everything handed on is known to sit on the SAME XCD as the
// reader (batch_step.hip with a batch that is a multiple of 8) and every address handed on is written ONCE per launch,
// before anything in the launch reads it.  The XCD's L2 is then the point of coherence and no line of the data can be stale
// in the reader's L1 (invalidated at kernel start, never filled since), so the waits carry NO acquire fence -- an
// agent-scope acquire (buffer_inv sc1) costs 1.7 us and more per wave that issues it (MI355X_MICROARCH.md).
// (A scalar poll -- s_load_dword glc, served by the L2 without queueing behind the co-resident tile's vector loads -- was
// measured too: no faster at 64 matrices, 12 % slower at 8, where tiles spin on their inputs: 3.73 -> 4.18 ms.)
template <bool LOCALP>
__device__ __forceinline__ int poll_word(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool LOCALP>
__device__ __forceinline__ void acquire_unless_local() {
    if constexpr (!LOCALP) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else asm volatile("" ::: "memory");                      // (the compiler keeps later loads behind the poll all the same)
}
struct Chase {
    const int* p0 = nullptr;     // progress word of the X operand's source
    const int* p1 = nullptr;     // ... of the Z operand's
    int base0 = 0, base1 = 0;    // the word's value when block 0 of this tile's K range is NOT yet there
};
template <bool LOCALP = false>
__device__ __forceinline__ int chase_poll(const Chase& ch) {
    const int v0 = poll_word<LOCALP>(ch.p0) - ch.base0;
    const int v1 = poll_word<LOCALP>(ch.p1) - ch.base1;
    return __builtin_amdgcn_readfirstlane(v0 < v1 ? v0 : v1);
}
// The two polls issued EARLY (their results are not waited for here): a tile puts them ahead of the loads of its input
// tile, so that one memory round trip covers both; chase_wait_pre then starts from what they brought.
struct ChasePre { int v0, v1; };
template <bool LOCALP = false>
__device__ __forceinline__ ChasePre chase_issue(const Chase& ch) {
    ChasePre p;
    p.v0 = poll_word<LOCALP>(ch.p0);
    p.v1 = poll_word<LOCALP>(ch.p1);
    return p;
}
// leading K blocks of the tile that are complete: >= need on return, or the last value seen after a time-out (ok = false)
template <bool LOCALP = false>
__device__ __forceinline__ int chase_wait(const Chase& ch, int need, bool& ok) {
    int r = chase_poll<LOCALP>(ch);
    if (r < need) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        while ((r = chase_poll<LOCALP>(ch)) < need) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > CHASE_LIMIT_TICKS) {
                ok = false;
                break;
            }
        }
    }
    acquire_unless_local<LOCALP>();
    return r;
}
template <bool LOCALP = false>
__device__ __forceinline__ int chase_wait_pre(const Chase& ch, const ChasePre& pre, int need, bool& ok) {
    const int a = pre.v0 - ch.base0, b = pre.v1 - ch.base1;
    const int r = __builtin_amdgcn_readfirstlane(a < b ? a : b);
    if (r < need) return chase_wait<LOCALP>(ch, need, ok);
    acquire_unless_local<LOCALP>();
    return r;
}

template <int WL, bool CHASE = false, bool LOCALP = false>
__device__ __forceinline__ void gemm_nt_128(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                               int64_t ldb, int nchunks, f32x16 (&acc)[4], float* smem,
                                               const Chase* ch = nullptr, bool* ch_ok = nullptr) {
    if (nchunks <= 0) return;
    int ready = 0;                             // CHASE: leading K blocks known complete
    if constexpr (CHASE) ready = chase_wait<LOCALP>(*ch, 1, *ch_ok);
    StageRegs s0, s1;
    const StageAddr sa = stage_addr(A, lda, B, ldb);
    stage_load_buf(s0, sa, 0);                 // (nchunks >= 4: chunks 1 and 2 exist)
    stage_store(s0, smem);
    stage_load_buf(s0, sa, BK);
    stage_load_buf(s1, sa, 2 * BK);
    __syncthreads();
    Frag<WL> F0, F1;
    frag_load<WL>(F0, smem, 0);
    float* b0 = smem;
    float* b1 = smem + STAGE_FLOATS;
    // two-level summation: P takes the products of one segment (4 chunks = 128 of K) from zero, acc is the running sum.
    // nchunks is a multiple of 4 at every call site ((kb1 - kb0) * TS / BK).
    f32x16 P[4];
    P[3] = zero16c();
    int c = 0;
    for (; c + SEG_CHUNKS < nchunks; c += SEG_CHUNKS) {        // every segment but the last: all loads / stores / reads exist
        if constexpr (CHASE) {                 // this segment requests the next block's chunks
            const int need = (c >> 2) + 2;
            if (ready < need) ready = chase_wait<LOCALP>(*ch, need, *ch_ok);
        }
        chunk_run<WL, true, true, true, SEG_FIRST>(b0, b1, F0, F1, P, acc, s0, sa, (c + 3) * BK);
        chunk_run<WL, true, true, true>(b1, b0, F0, F1, P, acc, s1, sa, (c + 4) * BK);
        chunk_run<WL, true, true, true>(b0, b1, F0, F1, P, acc, s0, sa, (c + 5) * BK);
        chunk_run<WL, true, true, true, SEG_LAST>(b1, b0, F0, F1, P, acc, s1, sa, (c + 6) * BK);
    }
    // the last segment: chunk c+3 is the last to be requested, chunk c+3 the last to be stored
    chunk_run<WL, true, true, true, SEG_FIRST>(b0, b1, F0, F1, P, acc, s0, sa, (c + 3) * BK);
    chunk_run<WL, true, true, false>(b1, b0, F0, F1, P, acc, s1, sa, 0);
    chunk_run<WL, true, true, false>(b0, b1, F0, F1, P, acc, s0, sa, 0);
    chunk_run<WL, false, false, false, SEG_LAST>(b1, b0, F0, F1, P, acc, s1, sa, 0);
    acc[3] += P[3];                            // the last segment's last quarter
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
    Chase ch;                          // tri_tile_run<true> only: the progress words of X's and Z's sources
};

// One per-lane byte offset per operand (row t >> 3, 16-byte column t & 7); the row group p (32 rows further down) and
// the K offset of the chunk are wave-uniform and ride in the instruction's SGPR offset.
// Cache policy of the operand each tile reads ONCE (Z: its own block row): nt (aux = 2), so that it does not push the
// operand the tiles of a matrix SHARE (X: block row k) out of the XCD's L2 -- FETCH_SIZE per launch 1.95 -> 1.70 GB,
// speed unchanged (round 2 A/B; git history: scripts/archive/nt_exp.sh).
#ifndef VOLT_Z_AUX
#define VOLT_Z_AUX 2
#endif
struct TriSrc {
    __amdgpu_buffer_rsrc_t ra, rb, rw;
    int va, vb, vw;
    int pa, pb;                        // byte stride of a row group: 32 ldx * 4, 32 ldz * 4  (W: 32 * 128 * 4)
    int n1, nall;
};

// KIND (compile time): 1 = a phase-1 chunk (both operands), 2 = a W chunk (B operand only, chunk index c - n1), 0 = nothing;
// -1 = decided at run time from c (the prologue)
template <int KIND = -1>
__device__ __forceinline__ void tri_load_piece(StageRegs& s, const TriSrc& ts, int c, int p) {
    if constexpr (KIND == 0) return;
    if (KIND == 1 || (KIND < 0 && c < ts.n1)) {
        const int sa = __builtin_amdgcn_readfirstlane(c * BK * 4 + p * ts.pa);
        const int sb = __builtin_amdgcn_readfirstlane(c * BK * 4 + p * ts.pb);
        s.a[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ts.ra, ts.va, sa, 0));
        s.b[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ts.rb, ts.vb, sb, VOLT_Z_AUX));
    } else {
        const int so = __builtin_amdgcn_readfirstlane((c - ts.n1) * BK * 4 + p * (32 * TS * 4));
        s.b[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ts.rw, ts.vw, so, 0));
    }
}
template <int KIND = -1>
__device__ __forceinline__ void tri_store_piece(const StageRegs& s, float* __restrict__ buf, const TriSrc& ts, int c, int p) {
    if constexpr (KIND == 0) return;
    const int t = threadIdx.x;
    const int row = t >> 3, cq = (t & 7) * 4;
    if (KIND == 1 || (KIND < 0 && c < ts.n1)) *reinterpret_cast<f32x4*>(buf + (row + 32 * p) * SLD + cq) = s.a[p];
    *reinterpret_cast<f32x4*>(buf + TS * SLD + (row + 32 * p) * SLD + cq) = s.b[p];
}

// phase-1 chunk `c` living in `cur`: stores chunk c+1 (from `s`) into `nxt`, requests chunk c+3 into `s`
// MORE: the next chunk is a phase-1 chunk too (Frag<1> reads behind the barrier);  STK / LDK: what chunk c+1 (stored from
// `s` into `nxt`) and chunk c+3 (requested into `s`) are -- 1 a phase-1 chunk, 2 a W chunk, 0 nothing.  Compile-time, like
// chunk_run's flags: phase 1 is a whole number of segments, so every chunk knows where it stands relative to the W chunks.
template <bool MORE, int STK, int LDK, int ROLE = SEG_MID>
__device__ __forceinline__ void tri_chunk_p1(const float* cur, float* nxt, int c, Frag<1>& F0, Frag<1>& F1,
                                             f32x16 (&P)[4], f32x16 (&T)[4], StageRegs& s, const TriSrc& ts) {
    frag_load<1>(F1, cur, 1);
    VOLT_SB();
    if constexpr (ROLE == SEG_FIRST) seg_first_group<1>(F0, P, T);
    else frag_mma<1>(F0, P);
    VOLT_SB();
    frag_load<1>(F0, cur, 2);
    VOLT_SB();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        frag_mma_m<1>(F1, m, P);
        tri_store_piece<STK>(s, nxt, ts, c + 1, m);
        VOLT_SB();
    }
    frag_load<1>(F1, cur, 3);
    VOLT_SB();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        frag_mma_m<1>(F0, m, P);
        tri_load_piece<LDK>(s, ts, c + 3, m);
        VOLT_SB();
    }
    __syncthreads();
    if constexpr (MORE) frag_load<1>(F0, nxt, 0);
    VOLT_SB();
    if constexpr (ROLE == SEG_LAST) seg_last_group<1>(F1, P, T);
    else frag_mma<1>(F1, P);
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
        if (TP < 3) tri_store_piece<2>(s, nxt, ts, c + 1, m);
        VOLT_SB();
    }
    fragw_load(G1, cur, TP, 3);
    VOLT_SB();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        fragw_mma_m(G0, T[TP], TP, 2, m, O);
        if (TP == 0) tri_load_piece<2>(s, ts, c + 3, m);        // only chunk n1+3 is still to be requested
        VOLT_SB();
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m) fragw_mma_m(G1, T[TP], TP, 3, m, O);
    VOLT_SB();
}

// Hand-off waits are bounded by WALL CLOCK (s_memrealtime: a constant 100 MHz counter), not by an iteration count: a
// preempted or profiled run spins more often, not longer.  3 s -- only a bug or a wedged device gets there.
#ifndef VOLT_WAIT_TICKS
#define VOLT_WAIT_TICKS 300000000ull
#endif
constexpr unsigned long long WAIT_LIMIT_TICKS = VOLT_WAIT_TICKS;
// `want` = 0: wait for a non-zero word;  else wait for the word to EQUAL `want` (flags that carry the number of the step
// they belong to and are never cleared: small_step_kernel in chol.hip).
template <bool LOCALP = false>
__device__ __forceinline__ bool flag_is_set(const int* flag, int want) {
    const int v = poll_word<LOCALP>(flag);
    return want ? v == want : v != 0;
}
template <bool LOCALP = false>
__device__ __forceinline__ bool wait_flag(const int* flag, int want, int sleep) {
    if (flag_is_set<LOCALP>(flag, want)) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned spins = 0;
    while (!flag_is_set<LOCALP>(flag, want)) {
        if (sleep == 2) __builtin_amdgcn_s_sleep(2);
        else __builtin_amdgcn_s_sleep(4);
        if ((++spins & 1023u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS) return false;
    }
    return true;
}
__device__ __forceinline__ bool wait_nonzero(const int* flag, int sleep) { return wait_flag(flag, 0, sleep); }
template <bool LOCALP = false>
__device__ __forceinline__ bool flag_wait_one_lane(const int* flag, int want = 0) {
    bool ok = true;
    if (threadIdx.x == 0) {
        ok = wait_flag<LOCALP>(flag, want, 4);
        acquire_unless_local<LOCALP>();
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

// T holds T0 on entry; O (zeroed here) holds the product on exit.  Returns false if a wait timed out (the W flag: in
// thread 0;  CHASE: in every lane of the wave that gave up).
template <bool CHASE = false, bool LOCALP = false>
__device__ __forceinline__ bool tri_tile_run(const TriTile& t, f32x16 (&T)[4], f32x16 (&O)[4], float* smem,
                                             const ChasePre* pre = nullptr) {
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
    int ready = 0;                                   // CHASE: leading K blocks known complete
    if constexpr (CHASE) {
        if (t.n1 > 0) ready = pre ? chase_wait_pre<LOCALP>(t.ch, *pre, 1, ok) : chase_wait<LOCALP>(t.ch, 1, ok);
    }
    if (t.flag && t.n1 == 0) {                       // no phase 1 to hide behind: W is the first thing needed
        ok = flag_wait_one_lane<LOCALP>(t.flag, t.want);
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
        // of K) from zero; the segment is added to T in the shadow of its last MFMAs.  n1 is a multiple of 4.
        Frag<1> F0, F1;
        frag_load<1>(F0, b0, 0);
        O[3] = zero16c();
        for (; c + SEG_CHUNKS < t.n1; c += SEG_CHUNKS) {          // every segment but the last: its prefetches are phase-1 chunks too
            if constexpr (CHASE) {                                // this segment requests the next block's chunks
                const int need = (c >> 2) + 2;
                if (ready < need) ready = chase_wait<LOCALP>(t.ch, need, ok);
            }
            tri_chunk_p1<true, 1, 1, SEG_FIRST>(b0, b1, c, F0, F1, O, T, s0, ts);
            tri_chunk_p1<true, 1, 1>(b1, b0, c + 1, F0, F1, O, T, s1, ts);
            tri_chunk_p1<true, 1, 1>(b0, b1, c + 2, F0, F1, O, T, s0, ts);
            tri_chunk_p1<true, 1, 1, SEG_LAST>(b1, b0, c + 3, F0, F1, O, T, s1, ts);
        }
        // the last segment of phase 1: the W chunks (n1 .. n1+3) come into view
        if (t.flag) ok = flag_wait_one_lane<LOCALP>(t.flag, t.want) && ok;      // barriers below order the acquire
        tri_chunk_p1<true, 1, 1, SEG_FIRST>(b0, b1, c, F0, F1, O, T, s0, ts);
        tri_chunk_p1<true, 1, 2>(b1, b0, c + 1, F0, F1, O, T, s1, ts);
        tri_chunk_p1<true, 1, 2>(b0, b1, c + 2, F0, F1, O, T, s0, ts);
        tri_chunk_p1<false, 2, 2, SEG_LAST>(b1, b0, c + 3, F0, F1, O, T, s1, ts);
        c += SEG_CHUNKS;
        T[3] += O[3];                                // the last segment's last quarter
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
    f32x16 P[4];                                     // two-level summation (see SEG_CHUNKS)
    P[3] = zero16c();
    for (; c + SEG_CHUNKS < t.n1; c += SEG_CHUNKS) {
        tri_chunk_p1<true, 1, 1, SEG_FIRST>(b0, b1, c, F0, F1, P, T, s0, ts);
        tri_chunk_p1<true, 1, 1>(b1, b0, c + 1, F0, F1, P, T, s1, ts);
        tri_chunk_p1<true, 1, 1>(b0, b1, c + 2, F0, F1, P, T, s0, ts);
        tri_chunk_p1<true, 1, 1, SEG_LAST>(b1, b0, c + 3, F0, F1, P, T, s1, ts);
    }
    tri_chunk_p1<true, 1, 1, SEG_FIRST>(b0, b1, c, F0, F1, P, T, s0, ts);      // the last segment: nothing follows it
    tri_chunk_p1<true, 1, 0>(b1, b0, c + 1, F0, F1, P, T, s1, ts);
    tri_chunk_p1<true, 1, 0>(b0, b1, c + 2, F0, F1, P, T, s0, ts);
    tri_chunk_p1<false, 0, 0, SEG_LAST>(b1, b0, c + 3, F0, F1, P, T, s1, ts);
    T[3] += P[3];
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

// alpha = Y z, one 128-column block of Y at a time: lane l of a half-wave holds columns 4 (l % 32) .. + 3 of the block
// for ITS row (lanes 0..31 one row, lanes 32..63 another) and the matching four entries of z; returns the two rows' dot
// products (wave-uniform).  The ONE definition of this sum: the launch-per-column tail (mll.hip, y_times_z_kernel) and
// the one-launch batched step (batch_step.hip, alpha items) add the same partials in the same order, so alpha agrees
// bit for bit between the schedules.
__device__ __forceinline__ void rowpair_dot(const f32x4& y, const f32x4& z, float& lo, float& hi) {
    // explicit fma / mul: the contraction must not depend on what the compiler makes of the call site
    float p = __builtin_fmaf(y[0], z[0], y[1] * z[1]) + __builtin_fmaf(y[2], z[2], y[3] * z[3]);
    p = dpp_add<0xB1>(p);
    p = dpp_add<0x4E>(p);
    p = dpp_add<0x141>(p);
    p = dpp_add<0x140>(p);                  // every lane of a 16-lane row holds the row's sum
    const int pi = __float_as_int(p);
    lo = __int_as_float(__builtin_amdgcn_readlane(pi, 0)) + __int_as_float(__builtin_amdgcn_readlane(pi, 16));
    hi = __int_as_float(__builtin_amdgcn_readlane(pi, 32)) + __int_as_float(__builtin_amdgcn_readlane(pi, 48));
}

// ---- hand-offs of the one-launch batched steps (batch_step.hip, batch64_step.hip) ---------------------------------------
// thread 0 waits for *p >= want (p may be nullptr), then one agent-scope acquire; a barrier for the rest
template <bool LOCALP>
__device__ __forceinline__ bool word_ge(const int* p, int want) { return poll_word<LOCALP>(p) >= want; }
template <bool LOCALP>
__device__ __forceinline__ bool wait_word_ge(const int* p, int want) {
    if (word_ge<LOCALP>(p, want)) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned spins = 0;
    while (!word_ge<LOCALP>(p, want)) {
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 1023u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS) return false;
    }
    return true;
}
template <bool LOCALP>
__device__ __forceinline__ void batch_wait(const int* p0, int want0, const int* p1, int want1, int* info_b) {
    if (threadIdx.x == 0) {
        bool ok = true;
        if (p0) ok = wait_word_ge<LOCALP>(p0, want0);
        if (p1) ok = wait_word_ge<LOCALP>(p1, want1) && ok;
        acquire_unless_local<LOCALP>();
        if (!ok) atomicCAS(info_b, 0, (int)0x80000000);
    }
    __syncthreads();
}
// behind plain stores: drain, barrier, agent-scope release, the word.  LOCALP (readers on this XCD): the drain alone -- the
// stores are in this L2 -- and a plain store of the word, which keeps its line there for the polls
// pub_tid: the thread that fences and stores the word -- a wave the caller's latency chain does not wait for, if it has one
template <bool LOCALP>
__device__ __forceinline__ void batch_publish_release(int* word, int val, int pub_tid = 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if ((int)threadIdx.x == pub_tid) {
        if constexpr (LOCALP) {
            *reinterpret_cast<volatile int*>(word) = val;
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(word, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// behind written-through (sc1) stores: every storing wave drains, then the word -- nothing is left in L2 to write back
template <bool LOCALP>
__device__ __forceinline__ void batch_publish_wt(int* word, int val) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if constexpr (LOCALP) *reinterpret_cast<volatile int*>(word) = val;
        else __hip_atomic_store(word, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


// log of the product of up to 16 pivots L_ii (a thread's chunk of the diagonal): ONE double log instead of sixteen software
// logs -- but with a single pivot's semantics kept: any pivot <= 0 or NaN makes the result NaN (an even number
// of negative pivots must not give a finite, plausible log-determinant in the loops that read info late), and a product that
// left the comfortable double range (sixteen fp32 values can reach 1e+-600) is redone pivot by pivot.
__device__ __forceinline__ double log_pivot_product(const float (&dg)[16], int nvalid_mask) {
    double prod = 1.0;
    bool ok = true;
#pragma unroll
    for (int u = 0; u < 16; ++u)
        if ((nvalid_mask >> u) & 1) {
            ok = ok && (dg[u] > 0.f);                        // false for <= 0 and for NaN
            prod *= (double)dg[u];
        }
    if (!ok) return __builtin_nan("");
    if (prod > 1e-280 && prod < 1e280) return log(prod);
    double s = 0.0;                                          // (rare: extreme pivots, or +inf on the diagonal)
#pragma unroll 1
    for (int u = 0; u < 16; ++u)
        if ((nvalid_mask >> u) & 1) s += log((double)dg[u]);
    return s;
}

// ---- who runs which piece (round 6) ------------------------------------------------------------------------------------
// Round 5 ran piece w as workgroup w and leaned on two things HIP does not promise (MI355X_MICROARCH.md, "Workgroup
// dispatch, XCD placement"): workgroups start in grid order (deadlock freedom) and workgroup w sits on XCD w % 8 (LOCAL's
// fence-free hand-offs through one L2).  Now the grid is a set of PULLERS -- as many workgroups as the chip holds at once,
// though nothing depends on that number -- and a piece is whatever the next ticket of a queue says:
//   * a waiter only ever waits for a piece with a SMALLER ticket of its own queue (the list is topologically ordered), and
//     a ticket is taken by a workgroup that is running: whatever is waited for is running or finished, whatever order
//     and wherever the dispatcher starts workgroups;
//   * LOCAL (batch a multiple of 8): eight queues, queue q = the pieces of the matrices b = q (mod 8) -- piece 8 t + q of
//     the list is ticket t of queue q, the list being matrix-innermost.  A workgroup asks the HARDWARE which XCD it is on
//     (s_getreg HW_REG_XCC_ID) and pulls from the queues that XCD owns; ownership is one compare-and-swap per queue
//     (claim[q] = XCD + 1): first its own number, and when that queue is dry, any queue nobody has claimed (an XCD that got
//     no workgroup at all -- a CU mask -- leaves an orphan that the others adopt whole).  So every piece of a matrix runs
//     under ONE L2 because the workgroups that run them read their own XCC_ID, not because of where workgroup w landed.
//   * otherwise: one queue, the agent-scope protocol, any workgroup anywhere.
// Queue words (ints, behind the progress words, cleared with them): head of queue q at [32 q], claim at [32 q + 1].
constexpr int BATCH_QWORDS = 8 * 32;

// HW_REG_XCC_ID (hwreg 20), bits 3:0: the XCD this wave runs on
__device__ __forceinline__ int hw_xcc_id() { return (int)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7; }

// Thread 0's view of the queues: which one it pulls from and how many it has tried to claim (see the block comment above).
//   xskew (tuning / tests): added to the hardware's XCC id -- the queues then sit on other XCDs than their numbers say;
//   xdrop (tuning / tests): bit x set = the workgroups on XCD x leave at once, as if a CU mask had emptied it -- their queues
//   are adopted by the others.
struct BatchPull {
    int q = -1, scan = 0;
};
template <bool LOCAL>
__device__ __forceinline__ int batch_next_piece(BatchPull& p, int* __restrict__ qw, int xcc, int per_queue) {
    for (;;) {
        if (p.q >= 0) {
            const int t = __hip_atomic_fetch_add(qw + 32 * p.q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < per_queue) return LOCAL ? 8 * t + p.q : t;
            p.q = -1;
        }
        if (!LOCAL) {
            if (p.scan) return -1;
            p.scan = 1;
            p.q = 0;
            continue;
        }
        while (p.q < 0 && p.scan < 8) {
            const int cand = (xcc + p.scan) & 7;
            ++p.scan;
            int seen = __hip_atomic_load(qw + 32 * cand + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen == 0) {
                int expect = 0;
                if (__hip_atomic_compare_exchange_strong(qw + 32 * cand + 1, &expect, xcc + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT))
                    seen = xcc + 1;
                else
                    seen = expect;
            }
            if (seen == xcc + 1) p.q = cand;
        }
        if (p.q < 0) return -1;
    }
}


}  // namespace volt
