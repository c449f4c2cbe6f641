// The one-launch steps for ONE or a FEW series: short series (N <= 1024, small_step_kernel) and one long
// series (long_step_kernel) -- every piece of the step a workgroup of one launch, started by flags in dispatch order.
//   reference: the training-loop body voltron/train_utils.py:243-254 at the reference's own sizes (ntrain = 400,
//   experiments/stocks/ForecastGenerator.py:53-91) and at BASELINE config 2 (one series, N = 4096).
// Split out of chol.hip in round 5 (no behaviour change); the tile bodies are tiles.h's, the batched one-launch step for
// MANY series is batch_step.hip.
#include "common.h"
#include "tiles.h"
#include "host.h"
#include "long_sched.h"
#include "../../include/volt_hip.h"
#include "../../include/volt_hip_tune.h"
#include <algorithm>
#include <array>
#include <map>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace volt {

// ----------------------------------------------------------------------------- short series: the whole step in ONE launch
// The reference's own sizes (ntrain = 400, experiments/stocks/ForecastGenerator.py:53-91) are n = 4 block columns: the
// launch-per-column sequence above is 11 launches (pad, prepare, clear, 5 factor / trtri launches, 3 tails), and every
// block column pays the chain  W_k -> panel tile (k+1,k) written out -> read back into the update of (k+1,k+1) -> pivots.
// Here every piece of a series is ONE workgroup of one launch, and a piece starts when the flags of what it reads say so:
//   D(0)     the first diagonal block, straight from K.
//   S(k)     k >= 1, the SPINE: everything block column k-1 still owes diagonal block k, without leaving the CU.
//            Ahead of W_{k-1}: the look-ahead part of A[k,k] (blocks m < k-1, parked in A) and the first phase of panel
//            tile (k,k-1).  When W_{k-1} appears: the W product -> L[k,k-1] (out to memory for the others, and into LDS),
//            A[k,k] -= L[k,k-1] L[k,k-1]^T from LDS into the pivot image, factor + invert -> W_k.
//   P(i,k)   i >= k+2: the other panel tiles (two-phase, as in factor_step_kernel).
//   T(i,j)   tiles of Y = L^-T with the z / Frobenius partials; T(i,i) copies W_i^T.
//   grid order (piece-major, series-minor):  for k = 0..n-1:  D(0) | S(k);  P(k+2..n-1, k);  U(k+1);  T(k-1, 0..k-1)
//                                            then T(n-1, 0..n-1).
//   S(k)    waits L[k,k-2], L[k-1,k-2] (ahead), W_{k-1};                  publishes L[k,k-1], then W_k
//   P(i,k)  waits L[i,k-1], L[k,k-1] (phase 1), W_k (phase 2);            publishes L[i,k]
//   T(i,j)  waits L[i,i-1], Y[i-1,j] (phase 1), W_i (phase 2);            publishes Y[i,j]
// Every piece depends only on pieces EARLIER in the grid, workgroups are dispatched in grid order, so whatever a
// resident workgroup waits for is resident or finished (the protocol of trsv.hip and of the W_k hand-off above).  The
// one exception is the tail: the n pieces of the last row of a series wait for each other -- they sit next to each other
// in the grid (series-major), fewer than an XCD has slots.
// With B a multiple of 8 a series' pieces all land on one XCD (w % 8 = b % 8).
// The flags are never cleared: a flag word holds the NUMBER of the step that set it.  hdr[3] counts the steps done on
// this workspace; a workgroup reads it on entry (E), waits for E + 1, publishes E + 1, and the last workgroup to leave
// the launch (hdr[4] counts them) stores E + 1 back -- so a replayed hipGraph needs no host-side argument to change,
// and no clearing launch precedes the step.  The state is written once by volt_mll_workspace_init_f32; the kernel
// checks its header and reports scratch that is not (or no longer) initialised as info = INT_MIN + 1.
// Tail (mll.hip's three tail kernels, same arithmetic in the same order): every T piece takes a ticket when its
// reductions are out; the n pieces of the last row wait for the full count, each sums the z-partials and takes every
// n-th group of four rows of alpha = Y z; the last of THEM to finish (a second ticket) writes the scalars.
constexpr int SMALL_MAGIC = 0x564f4c53;
constexpr int SMALL_HDR = 64;                  // ints ahead of the per-series blocks
struct SmallState {
    int* hdr;                                  // [0] magic [1] B [2] n [3] steps done [4] workgroups that have left the launch
    int* ser;                                  // per series, `stride` ints apart (a 128-byte line of its own or more):
    int stride;                                //   [0] T pieces that have delivered alpha's partial sums (running total)
                                               //   [4 ..) sf[n][4]: column slab of diagonal block k handed on (16-byte rows)
                                               //          rowc[n]: T pieces of row i whose z-partials are out (running total)
                                               //          wf[n]: W_k published   lf[n][n]: L[i,j] stored   yf[n][n]: Y tile stored
                                               //          uf[n]: look-ahead part of A[k,k] parked
    long long* stamps;                         // tuning only (volt_tune_small_stamps): 16 per workgroup, else nullptr
};
__host__ __device__ inline int small_stride(int n) { return (4 + 7 * n + 2 * n * n + 31) & ~31; }
#define LONG_STAMP(i) do { if (st.stamps && threadIdx.x == 0) st.stamps[(int64_t)pw * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SMALL_STAMP(i) do { if (st.stamps && threadIdx.x == 0) st.stamps[(int64_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
struct SmallTail {
    const float* resid;                        // [B,N]
    float* rpad;                               // [B,Np]  zero-padded copy, written by D(0)
    float* z;                                  // [B,Np]
    float* apad;                               // [B,Np]
    float* apart;                              // [B,n,Np]  alpha's partial sums: [i][128 j + c] from tile (i,j) of the inverse
    const float* sigma2;
    float jitter;
    float* out;                                // [B,8]
    float* alpha;                              // [B,N]
    int N;
};

// thread 0 polls with a growing pause (f1 may be nullptr), one agent-scope acquire, a barrier for the rest
__device__ __forceinline__ bool wait_flag_backoff(const int* flag, int want) {
    if (flag_is_set(flag, want)) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned spins = 0;
    while (!flag_is_set(flag, want)) {
        if (spins < 16) __builtin_amdgcn_s_sleep(2);
        else if (spins < 64) __builtin_amdgcn_s_sleep(8);
        else __builtin_amdgcn_s_sleep(24);
        if ((++spins & 1023u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS) return false;
    }
    return true;
}
__device__ __forceinline__ void small_wait(const int* f0, const int* f1, int want, int* info_b) {
    if (threadIdx.x == 0) {
        bool ok = wait_flag_backoff(f0, want);
        if (f1) ok = wait_flag_backoff(f1, want) && ok;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (!ok) atomicCAS(info_b, 0, (int)0x80000000);
    }
    __syncthreads();
}
__device__ __forceinline__ void small_publish(int* flag, int val) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave drains
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// A tile below diagonal block k, solved by SUBSTITUTION against the column slabs that block's workgroup hands on while
// it is still pivoting (diag_body<.., SLABS>) -- nothing here waits for the inverse W_k.  On entry T = -U^T, U the tile
// with every block column m < k already subtracted (tri_tile_run's accumulator layout: T[j] register q <-> column
// p = 32 j + accrow(q), lane % 32 <-> row c of this wave's 32 rows).  Per slab j, as soon as its flag is up:
//     U_j -= L'_m L_kk[j,m]^T     for the slabs m < j already here (their L_kk blocks came with THEIR flags: this
//                                 happens BEFORE flag j is up): A = rows of L_kk[j,m], B = L'_m
//     L'_j = U_j X_j^T            behind flag j: 16 MFMAs, A = T[j] straight from the registers, B = rows of X_j
//                                 (W_k's block (j,j))
// so that behind every flag, the last one included, only one 32^3 product is left.  Operands that every lane reads a row of (X_j, L_kk)
// come straight from memory as b128 loads; L'_j goes through this wave's own 32 rows of the LDS tile sL (row stride WLD)
// -- accumulator layout in, operand layout out -- which is also where the spine picks the finished tile up.  No barrier:
// a wave only ever reads the rows it wrote.  The tile goes out to memory slab by slab.
// MODE 0: a panel tile.  MODE 1: the spine's tile (k,k-1) -- `acc` holds -C, C the look-ahead part of A[k,k], in the
// 2x2-wave layout of gemm_nt_128<0>; behind every slab (one barrier: all four waves' rows of it are in the tile) it takes
// the rank-32 update L'_j L'_j^T in that pipeline's K order, so that after the last slab only a quarter of the product is
// left, and the result lands in the pivot image (lower triangle, zeros above) that diag_body works on.  MODE 2: a tile of
// Y = L^-T (the same right-hand product against W_i): the slabs' products are kept in O for the reductions.
// The four slab flags of a diagonal block sit in one aligned 16-byte word and go up in order: ONE load tells how many of
// them are up, so a tile that arrives late polls once, not once per slab (every poll is a round trip on the chain).
// Lane 0: returns how many leading flags equal `want` once that is more than j; -1 on a time-out.
__device__ __forceinline__ int wait_slab_flags(const int* slab, int j, int want) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, 16, 0x00020000);
    auto count = [&]() {
        asm volatile("" ::: "memory");
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, 0, 0, AUX_SC1);
        int c = 0;
        if ((int)v[0] == want) {
            c = 1;
            if ((int)v[1] == want) {
                c = 2;
                if ((int)v[2] == want) c = (int)v[3] == want ? 4 : 3;
            }
        }
        return c;
    };
    int c = count();
    if (c > j) return c;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned spins = 0;
    while ((c = count()) <= j) {
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 1023u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > WAIT_LIMIT_TICKS) return -1;
    }
    return c;
}

struct NoOp { __device__ __forceinline__ void operator()() const {} };
template <int MODE, class F = NoOp, int AUXL = 16 /* AUX_SC1 */>
__device__ __forceinline__ bool substitute_tile(f32x16 (&T)[4], const float* __restrict__ Lkk, int Np,
                                                const float* __restrict__ Wk, const int* slab, int want,
                                                float* __restrict__ out, float* sL, f32x16 (&X)[4],
                                                long long* stamps = nullptr, F before_first_rank = F(),
                                                int* handon = nullptr) {
#define SUB_STAMP(i) do { if (stamps && threadIdx.x == 0) stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    float* mine = sL + (wave * 32) * WLD;
    float* outw = out + (int64_t)(wave * 32) * Np;
    // handon (MODE 0, the split spine): the tile goes out written THROUGH, and behind every slab each wave counts itself
    // into handon[j] once its 32 rows of the slab are out -- the piece that takes the rank-32 updates reads them from there
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)outw, 0, 0x7fffffff, 0x00020000);
    bool ok = true;
    // X_j and the L_kk blocks were written through (sc1) ahead of their flag and are read with sc1 loads behind it: the
    // hand-off costs neither side an L2-wide write-back / invalidate (with 64 series in flight those were what the
    // pivot chains were waiting for)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)Wk, 0, TS * TS * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)Lkk, 0, 0x7fffffff, 0x00020000);
    // rank-32 update of the spine's accumulators with slab jj of the (whole) L tile
    int hand_pend = -1;
    auto hand_on = [&]() {
        if (MODE == 0 && handon && hand_pend >= 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(handon + hand_pend, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hand_pend = -1;
        }
    };
    auto rank32 = [&](int jj) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ko = 32 * jj + 8 * g + 4 * lh;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(sL + (wr * 64 + l31) * WLD + ko);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sL + (wc * 64 + l31) * WLD + ko);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(sL + (wc * 64 + 32 + l31) * WLD + ko);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(sL + (wr * 64 + 32 + l31) * WLD + ko);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                X[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], b0[m], X[0], 0, 0, 0);
                X[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], b1[m], X[1], 0, 0, 0);
                X[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], b0[m], X[2], 0, 0, 0);
                X[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], b1[m], X[3], 0, 0, 0);
            }
        }
    };
    f32x4 la[3][4];                                                // L_kk[j, m], m < j, for the slab after the one in hand
    int nready = 0;
    const bool vec_flags = (reinterpret_cast<uintptr_t>(slab) & 15) == 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // ---- between the flags: what the slabs already here owe slab j (their L_kk blocks came with THEIR flags), and
        // for the spine the rank-32 update of the slab before, whose MFMAs cover the latency of those loads
        if (j > 0) {
            if (MODE == 1 && j == 1) before_first_rank();          // (long series: the spine's accumulators are loaded as late as this)
            if (MODE == 1) {
                __syncthreads();                                   // all four waves' rows of slab j-1 are in the tile
                rank32(j - 1);
            }
#pragma unroll
            for (int m = 0; m < 3; ++m)
                if (m < j) {
                    f32x4 lb[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) lb[g] = *reinterpret_cast<const f32x4*>(mine + l31 * WLD + 32 * m + 8 * g + 4 * lh);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            T[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(la[m][g][e], lb[g][e], T[j], 0, 0, 0);
                }
        }
        hand_on();                                                 // (the previous slab's stores have had that long to drain)
        // ---- behind flag j: one product
        if (j >= nready) {                                         // (wave-uniform: flags 0 .. nready-1 have been seen up)
            if (vec_flags) {
                int c = 0;
                if (lane == 0) c = wait_slab_flags(slab, j, want);
                c = __builtin_amdgcn_readfirstlane(c);
                if (c < 0) { ok = false; c = 4; }
                nready = c;
            } else if (lane == 0) {
                ok = wait_flag(slab + j, want, 2) && ok;
            }
        }
        asm volatile("" ::: "memory");                             // the loads below stay below the poll
        SUB_STAMP(6 + j);
        f32x4 xb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            xb[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                wrs, ((32 * j + l31) * TS + 32 * j + 4 * lh) * 4, 32 * g, AUXL));
        // the rows of L_kk the NEXT slab's between-the-flags work reads came with this flag and the ones before it: asked
        // for now, their trip is covered by this slab's product and stores
        if (j < 3) {
#pragma unroll
            for (int m = 0; m < 3; ++m)
                if (m <= j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        la[m][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            lrs, (int)((((int64_t)(32 * (j + 1) + l31)) * Np + 32 * m + 4 * lh) * 4), 32 * g, AUXL));
                }
        }
        f32x16 O = zero16();
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) O = __builtin_amdgcn_mfma_f32_32x32x2f32(T[j][4 * g + e], xb[g][e], O, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = accrow(q, lane);
            mine[c * WLD + 32 * j + l31] = -O[q];
            if (MODE == 0 && handon)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-O[q]), ors, (int)(((int64_t)c * Np + 32 * j + l31) * 4), 0, AUX_SC1);
            else
                VOLT_OUT_STORE(outw + (int64_t)c * Np + 32 * j + l31, -O[q]);
        }
        if (MODE == 2) X[j] = O;
        wave_lds_fence();
        if (MODE == 0 && handon) hand_pend = j;                    // counted in behind the NEXT slab's between-the-flags work
    }
    hand_on();
    SUB_STAMP(10);
    if (MODE == 1) {
        __syncthreads();
        SUB_STAMP(11);
        rank32(3);
        __syncthreads();                                           // the image overlays the L tile
        SUB_STAMP(12);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int r = wr * 64 + tm * 32 + accrow(q, lane);
                    const int c = wc * 64 + tn * 32 + l31;
                    sL[r * DT + c] = (c <= r) ? -X[tm * 2 + tn][q] : 0.f;
                }
        SUB_STAMP(13);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's part of the tile is out (long since)
        SUB_STAMP(14);
    }
    return ok;               // meaningful in lane 0 of each wave
}

// Phase 1 of a two-phase tile in two parts: the K blocks but the last as soon as THEIR inputs are there (flags e0, e1),
// the last block -- whose operand is the tile the previous spine has only just handed on -- behind flags l0, l1.  That
// leaves one 128^3 product (6.8 us on a CU) between the hand-on and the first slab of the next diagonal block (~8 us).
__device__ __forceinline__ void phase1_two_parts(TriTile t, f32x16 (&T)[4], float* smem, const int* e0, const int* e1,
                                                 const int* l0, const int* l1, int want, int* info_b) {
    const int blocks = t.n1 / 4;
    if (blocks <= 0) return;
    if (blocks > 1) {
        small_wait(e0, e1, want, info_b);
        t.n1 = 4 * (blocks - 1);
        tri_phase1_only(t, T, smem);
    }
    small_wait(l0, l1, want, info_b);
    t.X += (int64_t)(blocks - 1) * TS;
    t.Z += (int64_t)(blocks - 1) * TS;
    t.n1 = 4;
    tri_phase1_only(t, T, smem);
}

// -C into the spine's accumulators: C = the look-ahead part of A[k,k] (parked in A by U(k); for k = 1 the caller's K)
__device__ __forceinline__ void spine_load_c(const float* __restrict__ A, int Np, int k, int b, const KSource& src,
                                             f32x16 (&acc)[4], bool from_k = false) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;
    const float* Ab = A + (int64_t)b * Np * Np;
    const bool usek = k == 1 || from_k;
    const float add = usek ? ((src.sigma2 ? src.sigma2[b] : 0.f) + src.jitter) : 0.f;
    const float* Kb = usek ? src.K + (int64_t)b * src.bsk : nullptr;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r = wr * 64 + tm * 32 + accrow(q, lane);
                const int c = wc * 64 + tn * 32 + l31;
                acc[tm * 2 + tn][q] = -input_elem(src, Kb, add, Ab, Np, usek, k * TS + r, k * TS + c);
            }
}

// ---- tail.  z_i (block i of z = L^-1 r) is the sum of the z-partials of the tiles of row i of the inverse, i.e. of
// pieces that sit next to each other in the grid and finish together: each of them waits for the row's count, adds the
// partials up (128 values, every piece for itself) and multiplies ITS tile -- still in registers -- into alpha's partial
// sum apart[i][block j] = Y[j,i] z_i.  No pass over Y: y_times_z_kernel's 640 KB stream becomes 16 wave reductions.
__device__ __forceinline__ void row_z(const TriReduce& red, const SmallTail& tl, int Np, int i, int j, int b, float* sz) {
    const int n = Np / TS, tid = threadIdx.x;
    if (tid < TS) {
        // eight loads in flight, added in the order of ever (jb ascending): one load at a time this was up to 32 round trips
        // behind the last tile of a row -- and so behind the last tile of the step
        float a = 0.f;
        const float* zp = red.zpart + (int64_t)b * n * Np + i * TS + tid;
        int jb = 0;
        for (; jb + 8 <= i + 1; jb += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = zp[(int64_t)(jb + u) * Np];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; jb <= i; ++jb) a += zp[(int64_t)jb * Np];
        sz[tid] = a;
        if (j == 0) tl.z[(int64_t)b * Np + i * TS + tid] = a;
    }
    __syncthreads();
}
// off-diagonal tile (i,j): Y[c][r] = -O[rb][q], c = 32 wave + accrow(q), r = 32 rb + lane % 32
__device__ __forceinline__ void alpha_part(const f32x16 (&O)[4], const SmallTail& tl, int Np, int i, int j, int b,
                                           const float* sz) {
    const int n = Np / TS, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31;
    float zr[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) zr[rb] = sz[rb * 32 + l31];
    float* dst = tl.apart + ((int64_t)b * n + i) * Np + j * TS + wave * 32;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        float v = -((O[0][q] * zr[0] + O[1][q] * zr[1]) + (O[2][q] * zr[2] + O[3][q] * zr[3]));
        v = dpp_add<0xB1>(v);
        v = dpp_add<0x4E>(v);
        v = dpp_add<0x141>(v);
        v = dpp_add<0x140>(v);                                     // every lane of a 16-lane row holds the row's sum
        const int vi = __float_as_int(v);
        const float lo = __int_as_float(__builtin_amdgcn_readlane(vi, 0)) + __int_as_float(__builtin_amdgcn_readlane(vi, 16));
        const float hi = __int_as_float(__builtin_amdgcn_readlane(vi, 32)) + __int_as_float(__builtin_amdgcn_readlane(vi, 48));
        if (lane == 0) {                                           // lanes 0..31 hold row accrow(q, 0), lanes 32..63 that + 4
            dst[(q & 3) + 8 * (q >> 2)] = lo;
            dst[(q & 3) + 8 * (q >> 2) + 4] = hi;
        }
    }
}

// Diagonal tile (i,i) of the inverse for the one-launch step: Y[i,i] = W_i^T out, its z-partial and Frobenius partial,
// then (behind the row's count) its share of alpha.  Two threads per row / column instead of trtri_diag_body's one, and
// the residual staged in LDS: this tile is the last piece of its row to start (it needs the WHOLE of W_i).
__device__ __forceinline__ void small_diag_tile(const float* __restrict__ Winv, float* __restrict__ Y, int Np, int i, int b,
                                                const TriReduce& red, const SmallTail& tl, int* rowc, int want,
                                                int* info_b, float* smem) {
    const int n = Np / TS, tid = threadIdx.x;
    const float* W = Winv + ((int64_t)b * n + i) * TS * TS;
    float* srv = smem + TS * WLD;                                   // residual block i, then z_i
    float* sfr = srv + TS;                                          // 256 Frobenius partials
    for (int e = tid; e < TS * TS / 4; e += NT) {
        const int r = e >> 5, c = (e & 31) * 4;
        *reinterpret_cast<f32x4*>(smem + r * WLD + c) = *reinterpret_cast<const f32x4*>(W + r * TS + c);
    }
    if (tid < TS) srv[tid] = red.rpad[(int64_t)b * Np + i * TS + tid];
    __syncthreads();
    {
        const int r = tid >> 1, h = tid & 1;                        // column r of Y = row r of W: entries c <= r, c = h, h + 2, ..
        float fz = 0.f, ff = 0.f, fz1 = 0.f, ff1 = 0.f;
        const int lim = red.N - i * TS;                             // columns c < lim are inside the matrix
        int c = h;
        for (; c + 2 <= r; c += 4) {                                // two independent chains per thread
            const float y0 = smem[r * WLD + c], y1 = smem[r * WLD + c + 2];
            fz += y0 * srv[c];
            fz1 += y1 * srv[c + 2];
            if (c < lim) ff += y0 * y0;
            if (c + 2 < lim) ff1 += y1 * y1;
        }
        if (c <= r) {
            const float y = smem[r * WLD + c];
            fz += y * srv[c];
            if (c < lim) ff += y * y;
        }
        fz += fz1;
        ff += ff1;
        fz += __shfl_xor(fz, 1);
        if (h == 0) red.zpart[((int64_t)b * n + i) * Np + i * TS + r] = fz;
        sfr[tid] = ff;
    }
    __syncthreads();
    if (tid < 64) {
        const float tot = wave_sum_f((sfr[tid] + sfr[tid + 64]) + (sfr[tid + 128] + sfr[tid + 192]));
        if (tid == 0) red.frob[(int64_t)b * (n * (n + 1) / 2) + i * (i + 1) / 2 + i] = tot;
    }
    // ---- the partials are out: the row's count, z_i, this tile's share of alpha:  apart[i][block i][c] = sum_{r >= c}
    // W[r][c] z_i[r].  The tile itself (64 KB of stores that nothing in this row waits for) follows in small_diag_tile_out.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(rowc, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    small_wait(rowc, nullptr, (int)((unsigned)want * (unsigned)(i + 1)), info_b);
    row_z(red, tl, Np, i, i, b, srv);
    {
        const int c = tid >> 1, h = tid & 1;
        float ap = 0.f, ap1 = 0.f;
        int r = c + h;
        for (; r + 2 < TS; r += 4) {
            ap += smem[r * WLD + c] * srv[r];
            ap1 += smem[(r + 2) * WLD + c] * srv[r + 2];
        }
        if (r < TS) ap += smem[r * WLD + c] * srv[r];
        ap += ap1;
        ap += __shfl_xor(ap, 1);
        if (h == 0) tl.apart[((int64_t)b * n + i) * Np + i * TS + c] = ap;
    }
}
// ... and the tile, out of the W image that small_diag_tile left in LDS: Y[i,i] = W_i^T
__device__ __forceinline__ void small_diag_tile_out(float* __restrict__ Y, int Np, int i, int b, int* yflag, int want,
                                                    const float* smem) {
    float* Yd = Y + (int64_t)b * Np * Np + (int64_t)i * TS * Np + (int64_t)i * TS;
    for (int e = threadIdx.x; e < TS * TS; e += NT) {
        const int c = e >> 7, r = e & 127;                          // Y row c, column r
        Yd[(int64_t)c * Np + r] = (r >= c) ? smem[r * WLD + c] : 0.f;
    }
    small_publish(yflag, want);
}
constexpr int SMALL_SPARE = TS * WLD + TS + NT;                      // floats of the staging area small_diag_tile leaves alone

// The last T piece of the series to deliver: alpha = sum of the partial sums, the scalars (= mll_scalars_kernel).
// sred: 16 doubles of LDS that nothing else is using (the diagonal tile's W image is still wanted).
__device__ __forceinline__ void small_tail_scalars(const float* __restrict__ A, int Np, int b, const TriReduce& red,
                                                   const SmallTail& tl, double* sred) {
    const int n = Np / TS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* Ab = A + (int64_t)b * Np * Np;
    const int N = tl.N;
    double v[4] = {0, 0, 0, 0};                                     // z'z, sum log L_ii, alpha'alpha, tr K^-1
    // Sixteen columns per thread AT ONCE (c = c0 + 256 u): their partial sums are added in the same order as ever -- block row
    // i ascending -- but the loads of one block row go out together, and four block rows per trip: this is one workgroup's
    // pass over n x Np partials and Np strided diagonal entries behind the LAST tile of the step, and with one column at a
    // time -- or with a branch per column: 512 loads one after the other -- it was 76 us of a 1.19 ms step (1 x 4096); 11 now.
    constexpr int CPT = 16;
    for (int c0 = tid; c0 < Np; c0 += CPT * NT) {
        float al[CPT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) al[u] = 0.f;
#pragma unroll 2
        for (int i = 0; i < n; ++i) {
            const float* row = tl.apart + ((int64_t)b * n + i) * Np;
            float x[CPT];
#pragma unroll
            for (int u = 0; u < CPT; ++u) {                        // unconditional loads (clamped), selected below: no branch
                const int c = c0 + u * NT;                         // per column, sixteen loads in flight
                x[u] = row[c < Np ? c : Np - 1];
            }
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                const int c = c0 + u * NT;
                al[u] += (c < Np && i >= c / TS) ? x[u] : 0.f;
            }
        }
        float dg[CPT], zz[CPT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int c = c0 + u * NT;
            dg[u] = 1.f;
            zz[u] = 0.f;
            if (c < Np) tl.apad[(int64_t)b * Np + c] = al[u];
            if (c < N) {
                dg[u] = Ab[(int64_t)c * Np + c];
                zz[u] = tl.z[(int64_t)b * Np + c];
            }
        }
        int valid = 0;                                             // ONE double log per thread for its sixteen pivots: their product
#pragma unroll                                                     // (common.h, log_pivot_product: NaN for a failed pivot) -- sixteen
        for (int u = 0; u < CPT; ++u) {                            // software logs were 10 us behind the last tile
            const int c = c0 + u * NT;
            if (c < N) {
                const double zi = zz[u];
                v[0] += zi * zi;
                valid |= 1 << u;
                v[2] += (double)al[u] * al[u];
                tl.alpha[(int64_t)b * N + c] = al[u];
            }
        }
        v[1] += log_pivot_product(dg, valid);
    }
    const int nt = n * (n + 1) / 2;
    for (int i = tid; i < nt; i += NT) v[3] += red.frob[(int64_t)b * nt + i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
        if (lane == 0) sred[4 * wave + k] = v[k];
    }
    __syncthreads();
    if (tid == 0) {
        const double q = (sred[0] + sred[4]) + (sred[8] + sred[12]);
        const double ld = 2.0 * ((sred[1] + sred[5]) + (sred[9] + sred[13]));
        const double aa = (sred[2] + sred[6]) + (sred[10] + sred[14]);
        const double tr = (sred[3] + sred[7]) + (sred[11] + sred[15]);
        const double LOG_2PI = 1.8378770664093453;
        float* o = tl.out + (int64_t)b * 8;
        o[0] = (float)(-0.5 * (q + ld + N * LOG_2PI) / N);
        o[1] = (float)(0.5 * (aa - tr) / N);
        o[2] = (float)q;
        o[3] = (float)ld;
        o[4] = (float)tr;
        o[5] = (float)aa;
        o[6] = (tl.sigma2 ? tl.sigma2[b] : 0.f) + tl.jitter;
        o[7] = 0.f;
    }
}

__global__ __launch_bounds__(256, 2) void small_step_kernel(float* __restrict__ A, float* __restrict__ Winv,
                                                           float* __restrict__ Y, int* __restrict__ info, int Np, int B,
                                                           KSource src, TriReduce red, SmallState st, SmallTail tl, int tickets) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    __shared__ int s_last;
    const int n = Np / TS, tid = threadIdx.x;
    if (st.hdr[0] != SMALL_MAGIC || st.hdr[1] != B || st.hdr[2] != n) {          // not (or no longer) what init wrote
        if (tid == 0 && (int)blockIdx.x < B) info[blockIdx.x] = (int)0x80000001;
        return;
    }
    const int want = __hip_atomic_load(st.hdr + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    // ---- which piece.  Group k: D(0) | S(k), the panel pieces P(k+2.., k), U(k+1) -- piece-major, series-minor -- and then
    // the k tiles of row k-1 of the inverse, series-major (the pieces of a row next to each other: they wait for each other).
    // Which piece this workgroup runs: its grid index while the whole grid is resident at once (nothing then depends on the
    // order workgroups start in), a TICKET otherwise (round 6): a piece only waits for pieces listed before it, and a ticket is
    // taken by a running workgroup -- the step does not lean on workgroups being dispatched in grid order.  The ticket word
    // (hdr[32]) is never cleared: step number `want` starts at (want - 1) * gridDim.x.
    int w = blockIdx.x;
    if (tickets) {
        if (tid == 0) s_last = (int)((unsigned)__hip_atomic_fetch_add(st.hdr + 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                                     (unsigned)(want - 1) * gridDim.x);
        __syncthreads();
        w = __builtin_amdgcn_readfirstlane(s_last);
    }
    int b = 0, k = 0, i = 0, j = 0, kind = -1;                          // kind 0: D(0) / S(k)  1: P(i,k)  2: T(i,j)  3: U(k)
    for (int kk = 0; kk <= n && kind < 0; ++kk) {
        const int np = kk < n && n - kk - 2 > 0 ? n - kk - 2 : 0;
        const int nu = (kk >= 1 && kk + 1 <= n - 1) ? 1 : 0;
        const int nh = kk < n ? 1 + np + nu : 0;                       // pieces ahead of the T pieces
        if (w < nh * B) {
            const int p = w / B;
            b = w % B;
            k = kk;
            if (p == 0) kind = 0;
            else if (p <= np) { kind = 1; i = kk + 1 + p; }
            else { kind = 3; k = kk + 1; }
        } else if (w < (nh + kk) * B) {
            w -= nh * B;
            kind = 2;
            i = kk - 1;
            if (B * n <= 224) {              // a whole row's pieces of every series fit the resident set: keep series-minor,
                j = w / B;                   // so that with B a multiple of 8 a series stays on its XCD (8 x 399: 139 -> 132 us)
                b = w % B;
            } else {                         // else the pieces of a row next to each other (they wait for each other)
                b = w / kk;
                j = w % kk;
            }
        } else {
            w -= (nh + kk) * B;
        }
    }
    int* ser = st.ser + (int64_t)b * st.stride;
    int* sf = ser + 4;
    int* rowc = sf + 4 * n;
    int* wf = rowc + n;
    int* lf = wf + n;
    int* yf = lf + n * n;
    int* uf = yf + n * n;
    int* info_b = info + b;
    float* Ab = A + (int64_t)b * Np * Np;
    SMALL_STAMP(0);

    if (kind == 0 && k == 0) {
        if (tid == 0) *info_b = 0;
        for (int c = tid; c < Np; c += NT) tl.rpad[(int64_t)b * Np + c] = c < tl.N ? tl.resid[(int64_t)b * tl.N + c] : 0.f;
        {   // the first diagonal tile straight from the caller's K into the pivot image: lower triangle only, every load
            // of a thread in flight at once (update_body's accumulator detour costs 5.3 us here, this 3)
            const float add = (src.sigma2 ? src.sigma2[b] : 0.f) + src.jitter;
            const float* Kb = src.K + (int64_t)b * src.bsk;
            float v[TS * TS / NT];
#pragma unroll
            for (int it = 0; it < TS * TS / 4 / NT; ++it) {
                const int e = tid + it * NT, r = e >> 5, c = (e & 31) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    v[4 * it + q] = (c + q <= r && r < src.N) ? Kb[(int64_t)r * src.ldk + c + q] : 0.f;
            }
#pragma unroll
            for (int it = 0; it < TS * TS / 4 / NT; ++it) {
                const int e = tid + it * NT, r = e >> 5, c = (e & 31) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x = v[4 * it + q];
                    if (c + q == r) x = r < src.N ? x + add : 1.f;
                    smem[r * DT + c + q] = x;
                }
            }
        }
        SMALL_STAMP(3);
        diag_body<false, true>(A, Winv, info, Np, 0, b, smem, nullptr, true, wf, want, sf);
        SMALL_STAMP(4);
    } else if (kind == 3) {
        // ---- U(k): what block columns m < k-1 owe the diagonal tile (k,k), parked in A for the spine
        small_wait(lf + k * n + (k - 2), nullptr, want, info_b);
        SMALL_STAMP(1);
        update_body<true>(A, Np, k, k, 0, k - 1, true, b, src, smem);
        SMALL_STAMP(3);
        small_publish(uf + k, want);
        SMALL_STAMP(4);
    } else if (kind <= 1) {
        // ---- ahead of diagonal block kd (spine: k - 1, panel piece: k): everything the earlier block columns owe this tile
        const int kd = kind == 0 ? k - 1 : k;                      // the diagonal block this tile sits under
        const int ti = kind == 0 ? k : i;                          // its block row
        SMALL_STAMP(1);
        TriJob jb = panel_job<true>(A, Winv, Np, ti, kd, b, src);
        f32x16 T[4];
        job_t0(jb, T);
        phase1_two_parts(jb.t, T, smem, lf + ti * n + (kd - 2), lf + kd * n + (kd - 2), lf + ti * n + (kd - 1),
                         lf + kd * n + (kd - 1), want, info_b);
        SMALL_STAMP(2);
        // ---- the chain: slab by slab behind the pivots of block kd
        f32x16 X[4];
        const float* Lkk = Ab + (int64_t)kd * TS * Np + (int64_t)kd * TS;
        const float* Wk = Winv + ((int64_t)b * n + kd) * TS * TS;
        bool ok;
        if (kind == 0) {
            if (k >= 2) small_wait(uf + k, nullptr, want, info_b);
            spine_load_c(A, Np, k, b, src, X);
            ok = substitute_tile<1>(T, Lkk, Np, Wk, sf + 4 * kd, want, jb.out, smem, X,
                                    st.stamps ? st.stamps + (int64_t)blockIdx.x * 16 : nullptr);
        } else {
            ok = substitute_tile<0>(T, Lkk, Np, Wk, sf + 4 * kd, want, jb.out, smem, X);
        }
        if ((tid & 63) == 0 && !ok) atomicCAS(info_b, 0, (int)0x80000000);
        SMALL_STAMP(3);
        if (kind == 0) {                                           // L[k,k-1] is handed on by a spare wave of diag_body
            SMALL_STAMP(4);
            diag_body<false, true>(A, Winv, info, Np, k, b, smem, nullptr, true, wf + k, want, sf + 4 * k, lf + k * n + kd);
        } else {
            small_publish(lf + ti * n + kd, want);
        }
    } else {
        if (i == j) {
            small_wait(wf + i, nullptr, want, info_b);
            SMALL_STAMP(1);
            small_diag_tile(Winv, Y, Np, i, b, red, tl, rowc + i, want, info_b, smem);
        } else {
            SMALL_STAMP(1);
            TriJob jb = trtri_job(A, Winv, Y, Np, i, j, b);
            f32x16 T[4], O[4];
            zero_acc(T);
            phase1_two_parts(jb.t, T, smem, lf + i * n + (i - 2), yf + (i - 2) * n + j, lf + i * n + (i - 1),
                             yf + (i - 1) * n + j, want, info_b);
            SMALL_STAMP(2);
            const bool ok = substitute_tile<2>(T, Ab + (int64_t)i * TS * Np + (int64_t)i * TS, Np,
                                               Winv + ((int64_t)b * n + i) * TS * TS, sf + 4 * i, want, jb.out, smem, O);
            if ((tid & 63) == 0 && !ok) atomicCAS(info_b, 0, (int)0x80000000);
            __syncthreads();                                       // the reductions' scratch overlays the waves' L rows
            trtri_reduce(O, Np, i, j, b, red, smem);
            // the tile and its partials are out; the row's count, z_i, this tile's share of alpha
            small_publish(yf + i * n + j, want);
            if (tid == 0) __hip_atomic_fetch_add(rowc + i, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            small_wait(rowc + i, nullptr, (int)((unsigned)want * (unsigned)(i + 1)), info_b);
            row_z(red, tl, Np, i, j, b, smem);
            alpha_part(O, tl, Np, i, j, b, smem);
        }
        SMALL_STAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this piece's share of alpha is out:
        __syncthreads();                                           // the last piece of the series to say so closes it
        SMALL_STAMP(4);
        if (tid == 0) {
            const int nT = n * (n + 1) / 2;
            const int t = __hip_atomic_fetch_add(ser, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (unsigned)(t + 1) == (unsigned)want * (unsigned)nT;
        }
        __syncthreads();
        static_assert((SMALL_SPARE % 2) == 0 && SMALL_SPARE + 32 <= 2 * STAGE_FLOATS, "16 doubles behind the diagonal tile's LDS");
        if (s_last) small_tail_scalars(A, Np, b, red, tl, reinterpret_cast<double*>(smem + SMALL_SPARE));
        if (i == j) small_diag_tile_out(Y, Np, i, b, yf + i * n + i, want, smem);
    }
    SMALL_STAMP(5);
    // ---- the last workgroup out closes the step
    if (tid == 0) {
        const int f = __hip_atomic_fetch_add(st.hdr + 4, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (f == (int)gridDim.x - 1) {
            __hip_atomic_store(st.hdr + 4, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st.hdr + 3, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void small_init_kernel(int* __restrict__ base, int count, int B, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    base[i] = i == 0 ? SMALL_MAGIC : i == 1 ? B : i == 2 ? n : 0;
}

// ----------------------------------------------------------------------------- ONE long series in one launch
// (long_sched.h) The pieces of small_step_kernel for 9 .. 32 block columns, the early part of every deep tile's first
// phase cut into K-slices that are pieces of their own.  Same flags, same step counter, same tail; one workgroup per CU.
struct LongState {
    int* hdr;                                  // as SmallState::hdr (B = 1)
    int* ser;                                  // [0] T pieces delivered  [4..) sf[n][4] rowc[n] wf[n] lf[n][n] yf[n][n] uf[n] ecnt[ncnt]
    const int4* items;                         // the piece list, one entry per workgroup
    const int4* uinfo;                         // [n] {_, slabs, first slab, counter} of the look-ahead tile U(k)
    float* eslab;                              // [nslabs][128*128] partial accumulators of the early-part slices
    long long* stamps;
    int xcd_from;                              // > 0: the spines S(g), g >= xcd_from, all run on XCD 0 (grid index % 8 == 0)
    int split;                                 // the plan has R(g) pieces: S(g) hands its tile on slab by slab
};
// acc += the nsl consecutive slabs at `slabs` (slab_dump's layout), read with sc1 loads: the slices wrote them through and
// raised a counter, no fence on either side
__device__ __forceinline__ void slab_add_sc1(f32x16 (&acc)[4], const float* __restrict__ slabs, int nsl) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slabs, 0, 0x7fffffff, 0x00020000);
    for (int sidx = 0; sidx < nsl; ++sidx) {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rs, ((t4 * 4 + g) * NT + (int)threadIdx.x) * 16, sidx * TS * TS * 4, AUX_SC1));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t4][4 * g + e] += v[e];
            }
    }
}
// a slice's partial sums out: write-through, every wave drained, then the tile's counter
__device__ __forceinline__ void slice_out(const f32x16 (&acc)[4], float* __restrict__ slab, int* counter) {
    slab_dump(acc, slab);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// rank-32 update of the diagonal tile's accumulators with columns [32 jj, 32 jj + 32) of the L tile in LDS (row stride WLD)
__device__ __forceinline__ void rank32_update(f32x16 (&X)[4], const float* sL, int jj) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ko = 32 * jj + 8 * g + 4 * lh;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(sL + (wr * 64 + l31) * WLD + ko);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sL + (wc * 64 + l31) * WLD + ko);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(sL + (wc * 64 + 32 + l31) * WLD + ko);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(sL + (wr * 64 + 32 + l31) * WLD + ko);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            X[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], b0[m], X[0], 0, 0, 0);
            X[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], b1[m], X[1], 0, 0, 0);
            X[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], b0[m], X[2], 0, 0, 0);
            X[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], b1[m], X[3], 0, 0, 0);
        }
    }
}

// LDS of the long-series step (namespace scope: the piece body below is a function of its own)
static __shared__ __attribute__((aligned(16))) float g_long_smem[2 * STAGE_FLOATS];
static __shared__ int g_long_last;
static __shared__ int g_long_piece;

// One piece of the long-series list.  pw: its index in the list (a TICKET since round 6: long_step_kernel below), want: the
// number of this step (the flags carry it), npieces: pieces of the list.
__device__ __forceinline__ void long_piece(float* __restrict__ A, float* __restrict__ Winv, float* __restrict__ Y,
                                           int* __restrict__ info, const int Np, const KSource src, const TriReduce red,
                                           const LongState st, const SmallTail tl, const int pw, const int want, const int npieces) {
    float* const smem = g_long_smem;
    int& s_last = g_long_last;
    const int n = Np / TS, tid = threadIdx.x;
    int4 item = st.items[pw];
    item.x = __builtin_amdgcn_readfirstlane(item.x);               // (uniform, said so here: batch_step.hip, batch_piece)
    item.y = __builtin_amdgcn_readfirstlane(item.y);
    item.z = __builtin_amdgcn_readfirstlane(item.z);
    item.w = __builtin_amdgcn_readfirstlane(item.w);
    const int pk = item.x & 255, pa = (item.x >> 8) & 255, pb = (item.x >> 16) & 255;
    const bool isE = pk >= LG_E_PANEL && pk <= LG_E_U;
    const int nslices = isE ? 0 : item.y;                           // base pieces: slices their early part came in
    const float* eslabs = st.eslab + (int64_t)item.z * TS * TS;
    int* ser = st.ser;
    int* sf = ser + 4;
    int* rowc = sf + 4 * n;
    int* wf = rowc + n;
    int* lf = wf + n;
    int* yf = lf + n * n;
    int* uf = yf + n * n;
    int* hf = uf + n;                                              // [n][4] waves of S(k) whose rows of slab j of L[k,k-1] are out (4 per step)
    int* ecnt = hf + 4 * n;
    int* info_b = info;
    float* Ab = A;
    LONG_STAMP(0);
    auto wait_slices = [&]() {
        if (tid == 0 && !wait_flag_backoff(ecnt + item.w, (int)((unsigned)want * (unsigned)nslices))) atomicCAS(info_b, 0, (int)0x80000000);
        __syncthreads();
    };

    if (pk == LG_D0) {
        if (tid == 0) *info_b = 0;
        for (int c = tid; c < Np; c += NT) tl.rpad[c] = c < tl.N ? tl.resid[c] : 0.f;
        update_body<true>(A, Np, 0, 0, 0, 0, true, 0, src, smem, true);
        diag_body<false, true>(A, Winv, info, Np, 0, 0, smem, nullptr, true, wf, want, sf);
    } else if (pk == LG_U || pk == LG_E_U) {
        // ---- the look-ahead part of diagonal tile (k,k): U(k) adds its early slices (if any) to its own block(s) and parks
        // C = input - sum in A for the spine; a slice E_U just dumps its partial sum
        const int kk = pa;
        f32x16 acc[4];
        zero_acc(acc);
        if (pk == LG_E_U) {                                        // (U itself multiplies nothing: all its blocks come as slabs, the
            const int u0 = item.y & 255, u1 = (item.y >> 8) & 255; //  last one from P(k,k-2) the moment that tile is there)
            small_wait(lf + kk * n + (u1 - 1), nullptr, want, info_b);
            const float* rows = Ab + (int64_t)kk * TS * Np + (int64_t)u0 * TS;
            gemm_nt_128<0>(rows, Np, rows, Np, (u1 - u0) * (TS / BK), acc, smem);
        }
        if (pk == LG_E_U) {
            slice_out(acc, st.eslab + (int64_t)item.z * TS * TS, ecnt + item.w);
        } else {
            // U(k): the early slabs first (they have been there for block columns), the input tile into registers, and then
            // the one slab that is only just being written: P(k,k-2)'s L[k,k-2] L[k,k-2]^T -- so that what the spine finds
            // parked in A is final and its own load is one round trip
            if (nslices > 0) {
                wait_slices();
                slab_add_sc1(acc, eslabs, nslices);
            }
            const int lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
            const float add = (src.sigma2 ? src.sigma2[0] : 0.f) + src.jitter;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int r = wr * 64 + tm * 32 + accrow(q, lane);
                        const int c = wc * 64 + tn * 32 + (lane & 31);
                        acc[tm * 2 + tn][q] -= input_elem(src, src.K, add, Ab, Np, true, kk * TS + r, kk * TS + c);
                    }
            const int4 ui = st.uinfo[kk];
            if (tid == 0 && !wait_flag_backoff(ecnt + ui.w, want)) atomicCAS(info_b, 0, (int)0x80000000);
            __syncthreads();
            slab_add_sc1(acc, st.eslab + (int64_t)ui.z * TS * TS, 1);
            float* C = Ab + (int64_t)kk * TS * Np + (int64_t)kk * TS;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int r = wr * 64 + tm * 32 + accrow(q, lane);
                        const int c = wc * 64 + tn * 32 + (lane & 31);
                        C[(int64_t)r * Np + c] = -acc[tm * 2 + tn][q];
                    }
            small_publish(uf + kk, want);
        }
    } else if (pk == LG_R) {
        // ---- the second half of a split spine: -A[k,k] (parked by U(k)) into the accumulators, the rank-32 update behind
        // every slab of L[k,k-1] that S(k) hands on, the pivot image, diagonal block k
        const int k = pa;
        f32x16 X[4];
        {
            const int4 ui = st.uinfo[k];                           // {U(k) exists, -, P(k,k-2)'s slab, its counter}
            if (ui.x) small_wait(uf + k, nullptr, want, info_b);
            spine_load_c(A, Np, k, 0, src, X, !ui.x);
            if (k == 2) {
                if (tid == 0 && !wait_flag_backoff(ecnt + ui.w, want)) atomicCAS(info_b, 0, (int)0x80000000);
                __syncthreads();
                slab_add_sc1(X, st.eslab + (int64_t)ui.z * TS * TS, 1);
            }
        }
        LONG_STAMP(1);
        const float* Lt = Ab + (int64_t)k * TS * Np + (int64_t)(k - 1) * TS;
        const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)Lt, 0, 0x7fffffff, 0x00020000);
        float* sL = smem;
        const int row = tid >> 1, half = tid & 1;
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            if (tid == 0 && !wait_flag(hf + 4 * k + j, 4 * want, 2)) atomicCAS(info_b, 0, (int)0x80000000);
            __syncthreads();
            if (st.stamps && tid == 0) st.stamps[(int64_t)pw * 16 + 6 + j] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    lrs, (int)(((int64_t)row * Np + 32 * j + 16 * half + 4 * g) * 4), 0, AUX_SC1));
                *reinterpret_cast<f32x4*>(sL + row * WLD + 32 * j + 16 * half + 4 * g) = v;
            }
            __syncthreads();
            rank32_update(X, sL, j);
        }
        __syncthreads();                                           // the image overlays the L tile
        {
            const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, wr = wave >> 1, wc = wave & 1;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int r = wr * 64 + tm * 32 + accrow(q, lane);
                        const int c = wc * 64 + tn * 32 + l31;
                        sL[r * DT + c] = (c <= r) ? -X[tm * 2 + tn][q] : 0.f;
                    }
        }
        LONG_STAMP(3);
        diag_body<false, true>(A, Winv, info, Np, k, 0, smem, nullptr, true, wf + k, want, sf + 4 * k);
        LONG_STAMP(4);
    } else if (pk == LG_TDIAG) {
        small_wait(wf + pa, nullptr, want, info_b);
        small_diag_tile(Winv, Y, Np, pa, 0, red, tl, rowc + pa, want, info_b, smem);
    } else {
        // ---- a two-phase tile: spine / panel tile under diagonal block kd, a tile of the inverse, or a K-slice of either.
        // ONE instance of the phase-1 pipeline serves them all: [early blocks], then [the last block] (a slice has only
        // the former, a tile whose early part came as slices only the latter).
        const bool isT = pk == LG_T || pk == LG_E_T;
        const int kd = pk == LG_SPINE ? pa - 1 : pb;               // S / P / E_PANEL: the diagonal block above the tile
        TriJob jb = isT ? trtri_job(A, Winv, Y, Np, pa, pb, 0) : panel_job<true>(A, Winv, Np, pa, kd, 0, src);
        const int blocks = jb.t.n1 / 4;                            // K blocks of the whole first phase
        int e0, e1, l0, l1;                                        // K blocks run here: the early range, the last block
        if (isE) {
            e0 = item.y & 255; e1 = (item.y >> 8) & 255;
            l0 = l1 = 0;
        } else {
            e0 = 0; e1 = nslices > 0 ? 0 : blocks - 1;
            l0 = blocks > 0 ? blocks - 1 : 0; l1 = blocks;
        }
        f32x16 T[4];
        if (isE || isT) zero_acc(T);
        else job_t0(jb, T);
#pragma unroll 1
        for (int part = 0; part < 2; ++part) {
            const int q0 = part == 0 ? e0 : l0, q1 = part == 0 ? e1 : l1;
            if (q1 <= q0) continue;
            // what a range ending at block q1 (exclusive) reads: S / P: L[row, q1-1] and L[kd, q1-1];  T(i,j): L[i, j+q1-1], Y[j+q1-1, j]
            small_wait(isT ? lf + pa * n + (pb + q1 - 1) : lf + pa * n + (q1 - 1),
                       isT ? yf + (pb + q1 - 1) * n + pb : lf + kd * n + (q1 - 1), want, info_b);
            TriTile t = jb.t;
            t.X += (int64_t)q0 * TS;
            t.Z += (int64_t)q0 * TS;
            t.n1 = 4 * (q1 - q0);
            tri_phase1_only(t, T, smem);
        }
        // the early part came as slices: added up BEHIND the last block -- the last slice can only start when the block column
        // before the last block's is complete and is the one input that may still be on its way when that block's operands
        // are there (added in front of it, the spine waited 2 - 3 us for it every column: 1 x 4096 1.215 -> 1.168 / 1.199 ms on two boxes)
        if (!isE && nslices > 0) {
            wait_slices();
            slab_add_sc1(T, eslabs, nslices);
        }
        LONG_STAMP(2);
        if (isE) {
            slice_out(T, st.eslab + (int64_t)item.z * TS * TS, ecnt + item.w);
        } else if (!isT) {
            f32x16 X[4];
            const float* Lkk = Ab + (int64_t)kd * TS * Np + (int64_t)kd * TS;
            const float* Wk = Winv + (int64_t)kd * TS * TS;
            bool ok;
            if (pk == LG_SPINE && st.split) {
                // split spine: the tile by substitution, handed on slab by slab to R(k); its flag goes up here
                ok = substitute_tile<0>(T, Lkk, Np, Wk, sf + 4 * kd, want, jb.out, smem, X,
                                        st.stamps ? st.stamps + (int64_t)pw * 16 : nullptr, NoOp(), hf + 4 * pa);
                if ((tid & 63) == 0 && !ok) atomicCAS(info_b, 0, (int)0x80000000);
                // every word of the tile went out written through and has been waited for (hand_on): the flag needs no
                // release fence -- an L2-wide write-back that would sit on the chain
                __syncthreads();
                if (tid == 0) __hip_atomic_store(lf + pa * n + kd, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (pk == LG_SPINE) {
                const int k = pa;
                // -A[k,k] (its look-ahead part, parked by U(k) -- whose last block is only a block column old) comes into the
                // accumulators right before the first rank-32 update, not before the first slab
                // -A[k,k] + what block columns m < k-1 owe it: the early blocks summed and parked by U(k), the last one a slab that
                // P(k,k-2) wrote the moment its tile existed
                auto load_c = [&]() {
                    const int4 ui = st.uinfo[k];                   // {U(k) exists, -, P(k,k-2)'s slab, its counter}
                    if (ui.x) small_wait(uf + k, nullptr, want, info_b);
                    spine_load_c(A, Np, k, 0, src, X, !ui.x);      // final, parked by U(k) -- or (k <= 2) the input tile itself ...
                    if (k == 2) {                                  // ... plus, for k = 2, the one slab there is
                        if (tid == 0 && !wait_flag_backoff(ecnt + ui.w, want)) atomicCAS(info_b, 0, (int)0x80000000);
                        __syncthreads();
                        slab_add_sc1(X, st.eslab + (int64_t)ui.z * TS * TS, 1);
                    }
                };
                load_c();
                // the slabs of diagonal block kd come from S(kd): when both spines run on XCD 0 (long_sched.h) its L2 has them --
                // written through it a moment ago -- and plain loads spare the trip through the fabric that sc1 loads make
                if (st.xcd_from > 0 && kd >= st.xcd_from)
                    ok = substitute_tile<1, NoOp, 0>(T, Lkk, Np, Wk, sf + 4 * kd, want, jb.out, smem, X,
                                                     st.stamps ? st.stamps + (int64_t)pw * 16 : nullptr);
                else
                    ok = substitute_tile<1>(T, Lkk, Np, Wk, sf + 4 * kd, want, jb.out, smem, X,
                                            st.stamps ? st.stamps + (int64_t)pw * 16 : nullptr);
                if ((tid & 63) == 0 && !ok) atomicCAS(info_b, 0, (int)0x80000000);
                LONG_STAMP(4);
                diag_body<false, true>(A, Winv, info, Np, k, 0, smem, nullptr, true, wf + k, want, sf + 4 * k, lf + k * n + kd);
            } else {
                ok = substitute_tile<0>(T, Lkk, Np, Wk, sf + 4 * kd, want, jb.out, smem, X);
                if ((tid & 63) == 0 && !ok) atomicCAS(info_b, 0, (int)0x80000000);
                if (pa == kd + 2) {
                    __syncthreads();                               // every wave's rows of the tile are in LDS
                    // L[k,k-2] is the last block diagonal tile k = pa is still owed, and it sits in this workgroup's LDS: its
                    // product with its own transpose goes out as the last of U(k)'s slabs right now (no round trip, no wait)
                    const int4 ui = st.uinfo[pa];
                    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5, wr = wave >> 1, wc = wave & 1;
                    zero_acc(X);
#pragma unroll 2
                    for (int kk = 0; kk < TS / 8; ++kk) {
                        const int ko = kk * 8 + 4 * lh;
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(smem + (wr * 64 + l31) * WLD + ko);
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(smem + (wc * 64 + l31) * WLD + ko);
                        const f32x4 b1 = *reinterpret_cast<const f32x4*>(smem + (wc * 64 + 32 + l31) * WLD + ko);
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(smem + (wr * 64 + 32 + l31) * WLD + ko);
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            X[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], b0[m], X[0], 0, 0, 0);
                            X[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], b1[m], X[1], 0, 0, 0);
                            X[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], b0[m], X[2], 0, 0, 0);
                            X[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], b1[m], X[3], 0, 0, 0);
                        }
                    }
                    slice_out(X, st.eslab + (int64_t)ui.z * TS * TS, ecnt + ui.w);
                }
                small_publish(lf + pa * n + kd, want);
            }
        } else {
            const int i = pa, j = pb;
            f32x16 O[4];
            const bool ok = substitute_tile<2>(T, Ab + (int64_t)i * TS * Np + (int64_t)i * TS, Np, Winv + (int64_t)i * TS * TS,
                                               sf + 4 * i, want, jb.out, smem, O);
            if ((tid & 63) == 0 && !ok) atomicCAS(info_b, 0, (int)0x80000000);
            __syncthreads();                                       // the reductions' scratch overlays the waves' L rows
            trtri_reduce(O, Np, i, j, 0, red, smem);
            small_publish(yf + i * n + j, want);
            if (tid == 0) __hip_atomic_fetch_add(rowc + i, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            small_wait(rowc + i, nullptr, (int)((unsigned)want * (unsigned)(i + 1)), info_b);
            row_z(red, tl, Np, i, j, 0, smem);
            alpha_part(O, tl, Np, i, j, 0, smem);
        }
    }
    if (pk == LG_T || pk == LG_TDIAG) {                            // the tiles of the inverse close the series' tail
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int nT = n * (n + 1) / 2;
            const int t = __hip_atomic_fetch_add(ser, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (unsigned)(t + 1) == (unsigned)want * (unsigned)nT;
        }
        __syncthreads();
        if (s_last) small_tail_scalars(A, Np, 0, red, tl, reinterpret_cast<double*>(smem + SMALL_SPARE));
        if (pk == LG_TDIAG) small_diag_tile_out(Y, Np, pa, 0, yf + pa * n + pa, want, smem);
    }
    LONG_STAMP(5);
    if (tid == 0) {                                                // the last workgroup out closes the step
        const int f = __hip_atomic_fetch_add(st.hdr + 4, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (f == npieces - 1) {
            __hip_atomic_store(st.hdr + 4, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st.hdr + 3, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


// The pieces are pulled BY TICKET by a grid of resident workgroups (round 6; the scheme of batch_step.hip / common.h, one queue):
// a piece only waits for pieces listed before it and a ticket is taken by a running workgroup, so the step no longer leans on
// workgroups being dispatched in grid order.  The ticket word (hdr[32], a line of its own) is never cleared: a step of G
// pullers takes exactly npieces + G tickets (every puller one too many), so step number `want` starts at
// (want - 1) * (npieces + G) -- like the flags, a replayed hipGraph needs no host-side argument to change.
// The loop is an irreducible cycle for the optimiser's sake (batch_step.hip).
__global__ __launch_bounds__(256, 1) void long_step_kernel(float* __restrict__ A, float* __restrict__ Winv,
                                                          float* __restrict__ Y, int* __restrict__ info, int Np,
                                                          KSource src, TriReduce red, LongState st, SmallTail tl, int npieces,
                                                          int never) {
    const int n = Np / TS, tid = threadIdx.x;
    if (st.hdr[0] != SMALL_MAGIC || st.hdr[1] != 1 || st.hdr[2] != n) {          // not (or no longer) what init wrote
        if (tid == 0 && blockIdx.x == 0) info[0] = (int)0x80000001;
        return;
    }
    const int want = __hip_atomic_load(st.hdr + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    const unsigned base = (unsigned)(want - 1) * ((unsigned)npieces + gridDim.x);
    int pw;
    if (never) {
        pw = never | (int)0x80000000;
        goto piece;
    }
pull_next:
    __syncthreads();                                               // the last piece's LDS traffic is over
    if (tid == 0) {
        const unsigned t = (unsigned)__hip_atomic_fetch_add(st.hdr + 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
        g_long_piece = t < (unsigned)npieces ? (int)t : -1;
    }
    __syncthreads();
    pw = g_long_piece;
piece:
    pw = __builtin_amdgcn_readfirstlane(pw);
    if (pw < 0) return;
    long_piece(A, Winv, Y, info, Np, src, red, st, tl, pw, want, npieces);
    goto pull_next;
}

}  // namespace volt

using namespace volt;

// ---- the one-launch step for short series (small_step_kernel): its state lives in the caller's workspace, written once
// by volt_mll_workspace_init_f32; like the schedule tables, a step uses it only for a region this library initialised.
static int small_pieces(int n) { return n * (n + 1) / 2 + (n - 1) * (n - 2) / 2 + n + (n > 2 ? n - 2 : 0); }   // workgroups per series
static bool small_applies(int B, int n) {
    const Tunables& tn = tunables();
    // measured (scripts/bench_small_step.py, profiles/r03/small_step_table.txt): the one launch wins while a series'
    // pieces find workgroup slots when their flags come up -- up to 40 series of 3 .. 4 block columns (16 of 8), 64 of one
    // or two; beyond that the pieces wait for slots rather than for each other and the launch-per-column path is faster
    if (n < 1 || n > tn.small_nmax || n > 8) return false;
    if (n <= 2) return B <= tn.small_maxb2;
    return B <= tn.small_maxb && (int64_t)B * small_pieces(n) <= tn.small_maxwg;
}
size_t volt_internal_small_bytes(int B, int n) {
    if (!small_applies(B, n)) return 0;
    return (((size_t)SMALL_HDR + (size_t)B * small_stride(n)) * sizeof(int) + 255) & ~(size_t)255;
}
static long long* g_small_stamps = nullptr;    // volt_tune_small_stamps
int volt_internal_small_install(void* state, size_t bytes, int B, int n, void* stream) {
    if (!state || !small_applies(B, n) || bytes < volt_internal_small_bytes(B, n)) return 0;
    const int count = SMALL_HDR + B * small_stride(n);
    hipLaunchKernelGGL(small_init_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, (int*)state, count, B, n);
    VOLT_LAUNCH_CHECK();
    return 0;
}
// 1: the step has been enqueued (one launch);  0: not applicable here (the caller runs the launch-per-column path)
int volt_internal_small_step(const float* K, int64_t ldk, int64_t bsk, const float* resid, const float* sigma2,
                             float jitter, float* A, float* Winv, float* Y, int* info, float* rpad, float* zpart,
                             float* frob, float* z, float* apad, float* apart, float* out, float* alpha, void* state,
                             int B, int N, void* stream) {
    const int Np = volt_padded_n(N), n = Np / TS;
    if (!state || !Y || !apart || !small_applies(B, n)) return 0;    // (state: only passed for a workspace declared initialised)
    int* base = (int*)state;
    const SmallState st{base, base + SMALL_HDR, small_stride(n), g_small_stamps};
    const KSource src{K, ldk, bsk, sigma2, jitter, N};
    const TriReduce red{rpad, zpart, frob, N};
    const SmallTail tl{resid, rpad, z, apad, apart, sigma2, jitter, out, alpha, N};
    // A pivot chain that shares its CU with another piece's MFMA / LDS traffic runs two to three times slower (64 x 399:
    // 71 us per diagonal block against 22): while every series can still have ~8 pieces resident, a workgroup gets a CU
    // to itself (16 KB of dynamic LDS on top of the 72 KB: one workgroup per 160 KB CU)
    const unsigned pad = B <= tunables().small_pad_maxb ? 16 * 1024 : 0;
    const int grid = B * small_pieces(n);
    const int tickets = grid > tunables().cus * (pad ? 1 : 2);     // more workgroups than the chip holds at once: pieces by ticket
    hipLaunchKernelGGL(small_step_kernel, dim3(grid), dim3(256), pad, (hipStream_t)stream, A, Winv, Y, info, Np,
                       B, src, red, st, tl, tickets);
    VOLT_LAUNCH_CHECK();
    return 1;
}

// ---- ONE long series in one launch (long_step_kernel, long_sched.h)
struct LongPlanDev {
    int4* items = nullptr;                     // pinned host
    int nitems = 0, nslabs = 0, ncnt = 0, xcd_from = 0;
};
// blocks in the slice next to the tile (measured with the split spine, 1 x 1500 ... 1 x 4096: 3 wins up to 24 block columns
// -- 1 x 2048 0.538 -> 0.527 ms --, 4 at 32 -- 1.235 -> 1.175)
static int long_first_for(int n) {
    const int f = tunables().long_first;
    return f > 0 ? f : (n <= 24 ? 3 : 4);
}
static const LongPlanDev* get_long_plan(int n, hipStream_t s) {
    static std::mutex mu;
    static std::map<std::array<int, 3>, LongPlanDev*> cache;
    const Tunables& tn = tunables();
    const int first = long_first_for(n);
    const std::array<int, 3> key{n, first, tn.long_emin};
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    const LongPlan pl = long_build(n, first, tn.long_emin, tn.long_xcd != 0, tn.long_split != 0);
    static_assert(sizeof(LongItem) == sizeof(int4), "items are read as int4");
    LongPlanDev* pd = new LongPlanDev;
    pd->nitems = (int)pl.items.size();
    pd->nslabs = pl.nslabs;
    pd->ncnt = pl.ncnt;
    pd->xcd_from = pl.xcd_from;
    if (hipHostMalloc((void**)&pd->items, (pl.items.size() + n) * sizeof(int4), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        delete pd;
        pd = nullptr;
    } else {
        memcpy(pd->items, pl.items.data(), pl.items.size() * sizeof(int4));
        memcpy(pd->items + pl.items.size(), pl.uinfo.data(), (size_t)n * sizeof(int4));
    }
    cache[key] = pd;
    return pd;
}
static bool long_applies(int B, int n) {
    const Tunables& tn = tunables();
    return tn.long_on && B == 1 && n > tn.long_nmin && n <= 32;     // (one series of 4 / 5 / 8 block columns: 0.133 / 0.168 / 0.257 ms here, 0.139 / 0.180 / 0.289 as a short series)
}
static size_t long_flag_ints(int n, int ncnt) { return (size_t)SMALL_HDR + (size_t)((4 + 11 * n + 2 * n * n + ncnt + 31) & ~31); }
// sizes of the plan for n block columns (the workspace layout asks for them on every step: computed once)
static void long_sizes(int n, size_t& items, int& nslabs, int& ncnt) {
    static std::mutex mu;
    static std::map<std::array<int, 3>, std::array<size_t, 3>> cache;
    const Tunables& tn = tunables();
    const int first = long_first_for(n);
    const std::array<int, 3> key{n, first, tn.long_emin};
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it == cache.end()) {
        const LongPlan pl = long_build(n, first, tn.long_emin, tn.long_xcd != 0, tn.long_split != 0);
        it = cache.emplace(key, std::array<size_t, 3>{pl.items.size(), (size_t)pl.nslabs, (size_t)pl.ncnt}).first;
    }
    items = it->second[0];
    nslabs = (int)it->second[1];
    ncnt = (int)it->second[2];
}
size_t volt_internal_long_bytes(int B, int n) {
    if (!long_applies(B, n)) return 0;
    size_t items;
    int nslabs, ncnt;
    long_sizes(n, items, nslabs, ncnt);
    return ((long_flag_ints(n, ncnt) * sizeof(int) + 255) & ~(size_t)255) + (((items + n) * sizeof(int4) + 255) & ~(size_t)255);
}
size_t volt_internal_long_slab_floats(int B, int n) {
    if (!long_applies(B, n)) return 0;
    size_t items;
    int nslabs, ncnt;
    long_sizes(n, items, nslabs, ncnt);
    return (size_t)nslabs * TS * TS;
}
int volt_internal_long_install(void* state, size_t bytes, int B, int n, void* stream) {
    if (!state || !long_applies(B, n) || bytes < volt_internal_long_bytes(B, n)) return 0;
    const LongPlanDev* pd = get_long_plan(n, (hipStream_t)stream);
    if (!pd) return 0;
    const int count = (int)long_flag_ints(n, pd->ncnt);
    hipLaunchKernelGGL(small_init_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, (int*)state, count, 1, n);
    VOLT_LAUNCH_CHECK();
    char* tab = reinterpret_cast<char*>(state) + (((size_t)count * sizeof(int) + 255) & ~(size_t)255);
    hipError_t e = hipMemcpyAsync(tab, pd->items, (size_t)(pd->nitems + n) * sizeof(int4), hipMemcpyHostToDevice, (hipStream_t)stream);
    return e != hipSuccess ? (int)e : 0;
}
// 1: enqueued (one launch);  0: not applicable (launch-per-column path)
int volt_internal_long_step(const float* K, int64_t ldk, int64_t bsk, const float* resid, const float* sigma2,
                            float jitter, float* A, float* Winv, float* Y, int* info, float* rpad, float* zpart,
                            float* frob, float* z, float* apad, float* apart, float* eslab, float* out, float* alpha,
                            void* state, int B, int N, void* stream) {
    const int Np = volt_padded_n(N), n = Np / TS;
    hipStream_t s = (hipStream_t)stream;
    if (!state || !Y || !apart || !eslab || !long_applies(B, n)) return 0;   // (state: only for a workspace declared initialised)
    const LongPlanDev* pd = get_long_plan(n, s);
    if (!pd) return 0;
    int* base = (int*)state;
    const size_t flag_bytes = (long_flag_ints(n, pd->ncnt) * sizeof(int) + 255) & ~(size_t)255;
    const KSource src{K, ldk, bsk, sigma2, jitter, N};
    const TriReduce red{rpad, zpart, frob, N};
    const SmallTail tl{resid, rpad, z, apad, apart, sigma2, jitter, out, alpha, N};
    const int4* tab = reinterpret_cast<const int4*>(reinterpret_cast<char*>(state) + flag_bytes);
    // (xcd_from: the spines-on-XCD-0 variant assumed the grid index -> XCD map; pieces are pulled by ticket now, so it is off)
    const LongState st{base, base + SMALL_HDR, tab, tab + pd->nitems, eslab, g_small_stamps, 0, tunables().long_split};
    // one workgroup per CU (16 KB of LDS padding): a pivot chain that shares its CU runs 1.5 - 3x slower
    const unsigned pad = tunables().long_pad ? 16 * 1024 : 0;
    // as many pullers as the chip holds at once (nothing depends on the number: the kernel derives its ticket base from it)
    const int pullers = std::min(pd->nitems, tunables().cus * (pad ? 1 : 2) * std::max(1, tunables().long_pullers));
    const int grid = tunables().long_pullers > 0 ? pullers : pd->nitems;
    hipLaunchKernelGGL(long_step_kernel, dim3(grid), dim3(256), pad, s, A, Winv, Y, info, Np, src, red, st, tl, pd->nitems, 0);
    VOLT_LAUNCH_CHECK();
    return 1;
}

extern "C" {

int volt_long_describe(int n, int first, int emin, int* items, int max_items, int* nslabs, int* ncnt) {
    if (n < 1 || n > 32) return -1;
    if (first < 0) return -2;
    if (emin < -1) return -3;
    if (first == 0) first = long_first_for(n);                      // 0 / -1: what the step itself uses
    if (emin == -1) emin = tunables().long_emin;
    const LongPlan pl = long_build(n, first, emin, tunables().long_xcd != 0, tunables().long_split != 0);
    if (nslabs) *nslabs = pl.nslabs;
    if (ncnt) *ncnt = pl.ncnt;
    if (items)
        for (int i = 0; i < (int)pl.items.size() && i < max_items; ++i) memcpy(items + 4 * i, &pl.items[i], sizeof(LongItem));
    return (int)pl.items.size();
}

int volt_tune_small_stamps(long long* stamps) {
    g_small_stamps = stamps;
    return 0;
}

}  // extern "C"
