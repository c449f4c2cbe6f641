// Batched blocked Cholesky, triangular solves and triangular inverse.
//   reference call sites: gpytorch MVN.log_prob -> torch.linalg.cholesky (via train_utils.py:249),
//   psd_safe_cholesky / torch.cholesky_solve at voltron/rollout_utils.py:35,36,44.
//
// Layout: A [B,Np,Np] row-major fp32, Np a multiple of 128, lower triangle referenced.  The matrices
// of a group advance together, a launch's grid is (tiles of the stage) x B, so the batch supplies the
// parallelism a single 4096^2 factorisation lacks in its late panels; the batch itself is cut into
// groups that run the same launch sequence on different streams (run_factor_groups).
//
// Left-looking by 128-wide block columns k = 0..n-1:
//   P1  panel update   A[i,k] -= sum_{m<k} L[i,m] L[k,m]^T   (i >= k)   fp32 MFMA, K = 128 k
//   P2  diagonal block L[k,k] = chol(A[k,k]),  W_k = L[k,k]^-1          one workgroup / matrix, LDS
//   P3  panel solve    L[i,k] = A[i,k] W_k^T                 (i >  k)   fp32 MFMA, K = 128
// Left-looking keeps the accumulator of a panel tile in registers across the whole K range, so
// each tile of L is written once (N^2/2 words) instead of read-modify-written n times as in a
// right-looking sweep; HBM traffic is the operand reads, N^3/(6*128) words per matrix.
// P1 and P2 (and a row of the triangular inverse) share ONE launch per block column
// (factor_step_kernel); P3 is the second.
#include "common.h"
#include "tiles.h"
#include "host.h"
#include "sched.h"
#include "long_sched.h"
#include "../../include/volt_hip.h"
#include "../../include/volt_hip_tune.h"
#include <algorithm>
#include <array>
#include <map>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <utility>
#include <vector>

namespace volt {

// ----------------------------------------------------------------------------- prepare
// A = tril-tiles(K) + (sigma2 + jitter) I, identity in the padding.  Tile (ti,tj), tj <= ti.
__global__ __launch_bounds__(256) void prepare_kernel(const float* __restrict__ K, int64_t ldk, int64_t bsk,
                                                      const float* __restrict__ sigma2, float jitter,
                                                      float* __restrict__ A, int N, int Np, int col0_only) {
    // linear lower-triangular tile index -> (ti, tj); col0_only: just the first block column
    int t = blockIdx.x;
    int ti, tj;
    if (col0_only) {
        ti = t;
        tj = 0;
    } else {
        ti = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        while (ti * (ti + 1) / 2 > t) --ti;
        tj = t - ti * (ti + 1) / 2;
    }
    const int b = blockIdx.y;
    const float add = (sigma2 ? sigma2[b] : 0.f) + jitter;
    const float* Kb = K + (int64_t)b * bsk;
    float* Ab = A + (int64_t)b * Np * Np;
    const int cq = (threadIdx.x & 31) * 4;
    const int r0 = threadIdx.x >> 5;
    const bool vec_ok = ((ldk & 3) == 0) && ((bsk & 3) == 0) && (((uintptr_t)K & 15) == 0);
#pragma unroll 4
    for (int rr = r0; rr < TS; rr += 8) {
        const int i = ti * TS + rr;
        const int j = tj * TS + cq;
        f32x4 v;
        if (i < N && j + 3 < N && vec_ok) {
            v = *reinterpret_cast<const f32x4*>(Kb + (int64_t)i * ldk + j);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (i < N && j + c < N) ? Kb[(int64_t)i * ldk + j + c] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (i == j + c) v[c] = (i < N) ? v[c] + add : 1.f;
        *reinterpret_cast<f32x4*>(Ab + (int64_t)i * Np + j) = v;
    }
}

template <bool FROMK>
__global__ __launch_bounds__(256, 2) void factor_step_kernel(float* __restrict__ A, float* __restrict__ Winv,
                                                            float* __restrict__ Y, int* __restrict__ info, int Np,
                                                            int k_upd, int i_tri, int B, KSource src, TriReduce red) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    static_assert(DIAG_LDS_FLOATS <= 2 * STAGE_FLOATS, "diag block must fit the staging area");
    static_assert(TS * WLD + TS <= 2 * STAGE_FLOATS, "W image + reduction scratch must fit");
    const int n = Np / TS;
    int w = blockIdx.x;
    TriJob jb;
    bool have = false;
    if (k_upd >= 0) {
        const int k = k_upd;
        if (w < B) {                                                      // diagonal tile of matrix w
            if (k >= 2) update_body<FROMK>(A, Np, k, k, k - 1, k, false, w, src, smem, true);
            else if (k == 1) update_body<FROMK>(A, Np, 1, 1, 0, 1, true, w, src, smem, true);
            diag_body(A, Winv, info, Np, k, w, smem, nullptr, k > 0);
            return;
        }
        w -= B;
        const int npre = (k >= 1 && k + 1 < n) ? B : 0;
        if (w < npre) {
            update_body<FROMK>(A, Np, k + 1, k + 1, 0, k, true, w, src, smem);
            return;
        }
        w -= npre;
        const int npan = (n - k - 1) * B;
        if (w < npan) {
            int t, b;
            decode_tile_batch(w, n - k - 1, B, t, b);
            jb = panel_job<FROMK>(A, Winv, Np, k + 1 + t, k, b, src);
            have = true;
        }
        w -= npan;
    }
    if (!have) {
        int j, b;
        decode_tile_batch(w, i_tri + 1, B, j, b);
        if (j == i_tri) {
            trtri_diag_body(Winv, Y, Np, i_tri, b, red, smem);
            return;
        }
        jb = trtri_job(A, Winv, Y, Np, i_tri, j, b);
    }
    // ---- the two-phase tile (panel: update + solve; trtri: off-diagonal tile of row i_tri)
    f32x16 T[4], O[4];
    if (jb.c0 && jb.row_ok && jb.vec_ok) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(jb.c0 + 32 * tm + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) T[tm][4 * g + e] = -v[e];
            }
    } else if (jb.c0 && jb.row_ok) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int q = 0; q < 16; ++q) T[tm][q] = -jb.c0[32 * tm + 8 * (q >> 2) + (q & 3)];
    } else {
        zero_acc(T);
    }
    const bool ok = tri_tile_run(jb.t, T, O, smem);
    if (threadIdx.x == 0 && !ok) atomicCAS(info + jb.b, 0, (int)0x80000000);   // hand-off timed out: internal error
    tri_store(O, jb.out, Np);
    if (jb.i >= 0 && red.rpad) trtri_reduce(O, Np, jb.i, jb.j, jb.b, red, smem);
}

// ----------------------------------------------------------------------------- small batches: split-K step kernel
// With fewer than ~16 matrices a launch has far fewer tiles than the chip has workgroup slots, and its duration is
// that of its LONGEST tile (K = 128 k) running on one CU: one series of N = 4096 spent 3.4 of its 4.5 ms that way.
// Here every long product is cut into S K-slices, one workgroup each; a slice dumps its partial accumulators (64 KB)
// into a slab, and the LAST slice of a tile to arrive (one atomic ticket per tile) adds the slabs up IN SLICE ORDER --
// the sum does not depend on who is last -- and carries on: the W product for panel / trtri tiles, the store for the
// diagonal look-ahead.  (A first attempt at this in round 1 used fp32 atomics on the panel update only and gained
// nothing: the trtri tiles of the same launch are just as long.  All three products are split here.)
constexpr int VOLT_SPLITK_SLABS = 64;    // the split-K slab of a workspace holds 64 (n+1) tiles of 128 x 128 floats
struct SplitK {
    float* slab;         // [tiles of one launch][S][128*128]
    int* count;          // [n launches][tiles of one launch] arrival counters, zeroed per call
    int S;               // slice slots per tile in the grid
    int L;               // K blocks (of 128) per slice: a tile of kb blocks is cut into min(S, ceil(kb / L)) slices
    int cap;             // tile-slab rows ((n+1) tiles each) this group may use: S * B <= cap
    int4* tab = nullptr; // caller scratch for the balanced schedule's item tables (mid-size batches), tab_bytes long
    size_t tab_bytes = 0;
};

__device__ __forceinline__ void slab_sum(f32x16 (&acc)[4], const float* __restrict__ slabs, int nsl) {
    zero_acc(acc);
    for (int sidx = 0; sidx < nsl; ++sidx) {
        const f32x4* in = reinterpret_cast<const f32x4*>(slabs + (int64_t)sidx * TS * TS);
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = in[(t4 * 4 + g) * NT + threadIdx.x];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t4][4 * g + e] += v[e];
            }
    }
}
// true in every thread of the LAST workgroup of the tile to get here; that workgroup may then read every slab
__device__ __forceinline__ bool splitk_arrive(int* counter, int nsl) {
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every wave's write-through slab stores have landed
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (old == nsl - 1);
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return s_last != 0;
}

// One K-slice [b0, b1) of the diagonal look-ahead for tile (k+1,k+1) of matrix b; the last of its nsl slices to arrive
// adds the slabs up and stores the tile.
template <bool FROMK>
__device__ __forceinline__ void lookahead_slice(float* __restrict__ A, int Np, int k, int b, int sl, int nsl, int b0, int b1,
                                                const KSource& src, float* __restrict__ slabs, int* __restrict__ counter,
                                                float* smem) {
    float* Ab = A + (int64_t)b * Np * Np;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
    f32x16 acc[4];
    zero_acc(acc);
    gemm_nt_128<0>(Ab + (int64_t)(k + 1) * TS * Np + (int64_t)b0 * TS, Np, Ab + (int64_t)(k + 1) * TS * Np + (int64_t)b0 * TS, Np,
                   (b1 - b0) * (TS / BK), acc, smem);
    if (nsl > 1) {
        slab_dump(acc, slabs + (int64_t)sl * TS * TS);
        if (!splitk_arrive(counter, nsl)) return;
        slab_sum(acc, slabs, nsl);
    }
    const bool usek = FROMK && src.K != nullptr;
    const float add = usek ? ((src.sigma2 ? src.sigma2[b] : 0.f) + src.jitter) : 0.f;
    const float* Kb = usek ? src.K + (int64_t)b * src.bsk : nullptr;
    float* C = Ab + (int64_t)(k + 1) * TS * Np + (int64_t)(k + 1) * TS;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r = wr * 64 + tm * 32 + accrow(q, lane);
                const int c = wc * 64 + tn * 32 + (lane & 31);
                C[(int64_t)r * Np + c] = input_elem(src, Kb, add, Ab, Np, usek, (k + 1) * TS + r, (k + 1) * TS + c)
                                         - acc[tm * 2 + tn][q];
            }
}

// One K-slice [b0, b1) of the first phase of a two-phase tile (panel or trtri); the last of its nsl slices to arrive
// adds the slabs up and carries on with the W product, the store and (trtri) the fused reductions.
template <bool FUSE>
__device__ __forceinline__ void tri_slice(TriJob& jb, int sl, int nsl, int b0, int b1, float* __restrict__ slabs,
                                          int* __restrict__ counter, int* __restrict__ info, const TriReduce& red, int Np,
                                          float* smem) {
    f32x16 T[4], O[4];
    if (sl == 0 && jb.c0 && jb.row_ok && jb.vec_ok) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(jb.c0 + 32 * tm + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) T[tm][4 * g + e] = -v[e];
            }
    } else if (sl == 0 && jb.c0 && jb.row_ok) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int q = 0; q < 16; ++q) T[tm][q] = -jb.c0[32 * tm + 8 * (q >> 2) + (q & 3)];
    } else {
        zero_acc(T);
    }
    if (!FUSE || nsl > 1) {
        gemm_nt_128<1>(jb.t.X + (int64_t)b0 * TS, jb.t.ldx, jb.t.Z + (int64_t)b0 * TS, jb.t.ldz, (b1 - b0) * (TS / BK), T, smem);
        if (nsl > 1) {
            slab_dump(T, slabs + (int64_t)sl * TS * TS);
            if (!splitk_arrive(counter, nsl)) return;
            slab_sum(T, slabs, nsl);
        }
        jb.t.n1 = 0;                                         // phase 1 is done: the W product alone
    }                                                        // (FUSE: an uncut tile runs both phases in one pipelined pass)
    const bool ok = tri_tile_run(jb.t, T, O, smem);
    if (threadIdx.x == 0 && !ok) atomicCAS(info + jb.b, 0, (int)0x80000000);
    tri_store(O, jb.out, Np);
    if (jb.i >= 0 && red.rpad) trtri_reduce(O, Np, jb.i, jb.j, jb.b, red, smem);
}

template <bool FROMK>
__global__ __launch_bounds__(256, 2) void factor_step_split_kernel(float* __restrict__ A, float* __restrict__ Winv,
                                                                  float* __restrict__ Y, int* __restrict__ info, int Np,
                                                                  int k_upd, int i_tri, int B, KSource src, TriReduce red,
                                                                  SplitK sk) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    const int n = Np / TS, S = sk.S;
    int w = blockIdx.x;
    const int k = k_upd;                                     // k >= 0 always here (the trailing trtri row passes k = n)
    // ---- diagonal tiles: as in factor_step_kernel
    if (k < n && w < B) {
        if (k >= 2) update_body<FROMK>(A, Np, k, k, k - 1, k, false, w, src, smem, true);
        else if (k == 1) update_body<FROMK>(A, Np, 1, 1, 0, 1, true, w, src, smem, true);
        diag_body(A, Winv, info, Np, k, w, smem, nullptr, k > 0);
        return;
    }
    if (k < n) w -= B;
    int* count = sk.count + (int64_t)k * B * (n + 1);        // this launch's counters
    const int npre = (k >= 1 && k + 1 < n) ? B : 0;
    // ---- diagonal look-ahead, split: tile index = matrix
    if (w < npre * S) {
        const int b = w / S, sl = w % S;
        const int kb = k;
        int nsl = (kb + sk.L - 1) / sk.L;
        nsl = nsl < 1 ? 1 : (nsl > S ? S : nsl);
        if (sl >= nsl) return;
        lookahead_slice<FROMK>(A, Np, k, b, sl, nsl, sl * kb / nsl, (sl + 1) * kb / nsl, src,
                               sk.slab + (int64_t)b * S * TS * TS, count + b, smem);
        return;
    }
    w -= npre * S;
    // ---- panel and trtri tiles, split: two-phase tiles whose first phase is cut into slices
    TriJob jb;
    int tile, kb;                                            // tile index within the launch (after the B look-ahead tiles), K blocks
    const int npan = (k < n) ? (n - k - 1) * B : 0;
    const int sl = w % S;
    int wt = w / S;
    if (wt < npan) {
        int t, b;
        decode_tile_batch(wt, n - k - 1, B, t, b);
        jb = panel_job<FROMK>(A, Winv, Np, k + 1 + t, k, b, src);
        tile = B + wt;
        kb = k;
    } else {
        wt -= npan;
        int j, b;
        decode_tile_batch(wt, i_tri + 1, B, j, b);
        if (j == i_tri) {
            if (sl == 0) trtri_diag_body(Winv, Y, Np, i_tri, b, red, smem);
            return;
        }
        jb = trtri_job(A, Winv, Y, Np, i_tri, j, b);
        tile = B + npan + wt;
        kb = i_tri - j;
    }
    int nsl = (kb + sk.L - 1) / sk.L;
    nsl = nsl < 1 ? 1 : (nsl > S ? S : nsl);
    if (sl >= nsl) return;
    tri_slice<false>(jb, sl, nsl, sl * kb / nsl, (sl + 1) * kb / nsl, sk.slab + (int64_t)tile * S * TS * TS, count + tile,
                     info, red, Np, smem);
}

// ----------------------------------------------------------------------------- mid-size batches: scheduled step kernel
// One workgroup per piece of the host's list (sched.h): a diagonal block, a K-slice of a look-ahead / panel / trtri
// tile, a trtri diagonal tile -- in grid order longest first, launched with dynamic LDS padding so that one workgroup
// fits a CU and the dispatcher does the list scheduling.  Slabs, tickets and the W_k hand-off are the split kernel's;
// the diagonal blocks lead the grid, so the panel tiles that poll their flags never wait for a workgroup that has
// not started.
template <bool FROMK>
__global__ __launch_bounds__(256, 1) void factor_step_sched_kernel(float* __restrict__ A, float* __restrict__ Winv,
                                                                  float* __restrict__ Y, int* __restrict__ info, int Np,
                                                                  int k, int i_tri, int B, KSource src, TriReduce red,
                                                                  SplitK sk, const int4* __restrict__ items, int4 key0,
                                                                  int4 key1) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    const int n = Np / TS, S = sk.S;
    {   // the caller's scratch must still hold the table this launch was sized for (volt_*_workspace_init): anything
        // else in there is reported, never followed
        const int4 h0 = sk.tab[0], h1 = sk.tab[1];
        if (h0.x != key0.x || h0.y != key0.y || h0.z != key0.z || h0.w != key0.w || h1.x != key1.x || h1.y != key1.y ||
            h1.z != key1.z || h1.w != key1.w) {
            if (blockIdx.x == 0)
                for (int b = threadIdx.x; b < B; b += NT) info[b] = (int)0x80000001;
            return;
        }
    }
    int* count = sk.count + (int64_t)k * B * (n + 1);        // this launch's counters
    const int4 d = items[blockIdx.x];
    const int kind = d.x & 7, b = d.x >> 3;
    const int sl = d.z & 255, nsl = (d.z >> 8) & 255, tile = d.z >> 16;
    const int b0 = d.w & 0xffff, b1 = d.w >> 16;
    float* slabs = sk.slab + (int64_t)tile * S * TS * TS;
    if (kind == 0) {
        if (k >= 2) update_body<FROMK>(A, Np, k, k, k - 1, k, false, b, src, smem, true);
        else if (k == 1) update_body<FROMK>(A, Np, 1, 1, 0, 1, true, b, src, smem, true);
        diag_body(A, Winv, info, Np, k, b, smem, nullptr, k > 0);
    } else if (kind == 4) {
        trtri_diag_body(Winv, Y, Np, i_tri, b, red, smem);
    } else if (kind == 1) {
        lookahead_slice<FROMK>(A, Np, k, b, sl, nsl, b0, b1, src, slabs, count + tile, smem);
    } else {
        TriJob jb = kind == 2 ? panel_job<FROMK>(A, Winv, Np, d.y, k, b, src) : trtri_job(A, Winv, Y, Np, i_tri, d.y, b);
        tri_slice<true>(jb, sl, nsl, b0, b1, slabs, count + tile, info, red, Np, smem);
    }
}

// Diagonal block alone with phase stamps (tuning hook): one workgroup per matrix on block column k of a COPY of A
__global__ __launch_bounds__(256, 2) void tune_diag_kernel(float* __restrict__ A, float* __restrict__ Winv,
                                                          int* __restrict__ info, int Np, int k, long long* stamps) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    diag_body<true>(A, Winv, info, Np, k, blockIdx.x, smem, stamps + 32 * blockIdx.x);
}

// What a factorisation needs cleared, in ONE launch: the first word of every W block (the "ready" flag of the block:
// diag_body / tri_tile_run), info, and the arrival counters of the split / scheduled launches when there are any.
__global__ void begin_factor_kernel(float* __restrict__ Winv, int nflags, int* __restrict__ info, int ninfo,
                                    int* __restrict__ count, int ncount) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nflags) reinterpret_cast<int*>(Winv)[(int64_t)i * TS * TS] = 0;
    if (i < ninfo) info[i] = 0;
    for (int c = i; c < ncount; c += gridDim.x * blockDim.x) count[c] = 0;
}

// Panel tiles alone (tuning hook: replayed on a finished factor -- W_k is there, the results are garbage)
__global__ __launch_bounds__(256, 2) void tune_update_kernel(float* __restrict__ A, const float* __restrict__ Winv,
                                                            int* __restrict__ info, int Np, int k, int B, KSource src) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    int t, b;
    decode_tile_batch(Np / TS - k - 1, B, t, b);
    const TriJob jb = panel_job<false>(A, Winv, Np, k + 1 + t, k, b, src);
    f32x16 T[4], O[4];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(jb.c0 + 32 * tm + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) T[tm][4 * g + e] = -v[e];
        }
    const bool ok = tri_tile_run(jb.t, T, O, smem);
    if (threadIdx.x == 0 && !ok) atomicCAS(info + jb.b, 0, (int)0x80000000);
    tri_store(O, jb.out, Np);
}
// The 2x2-wave update alone (the diagonal look-ahead's product), same grid.  ABL: ablations for the c0 breakdown.
template <int ABL>
__global__ __launch_bounds__(256, 2) void tune_update_sq_kernel(float* __restrict__ A, int Np, int k, int B, KSource src) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    int t, b;
    decode_tile_batch(Np / TS - k - 1, B, t, b);
    update_body<false, ABL>(A, Np, k + 1 + t, k, 0, k, true, b, src, smem);
}
__global__ void tune_empty_kernel(float* A) { if (A == nullptr) A[0] = 0.f; }

}  // namespace volt

using namespace volt;

// batch_step.hip: the whole factorisation in one launch
size_t volt_internal_batch_bytes(int B, int n, int has_y);
int volt_internal_batch_install(void* state, size_t bytes, int B, int n, int has_y, void* stream);
int volt_internal_batch_step(const float* K, int64_t ldk, int64_t bsk, const float* sigma2, float jitter, float* A,
                             float* Winv, float* Y, int* info, const float* rpad, float* zpart, float* frob, int B, int N,
                             float* z, float* apart, void* state, size_t state_bytes, void* stream, hipEvent_t e0,
                             hipEvent_t e1);

static Tunables read_env(Tunables t) {                       // VOLT_TUNE=1 processes only
    auto geti = [](const char* name, int& v) { if (const char* e = getenv(name)) v = atoi(e); };
        geti("VOLT_GROUPS", t.groups);
        geti("VOLT_SPLITK_TARGET", t.splitk_target);
        geti("VOLT_SPLITK_MINL", t.splitk_minl);
        geti("VOLT_SPLITK_MAXS", t.splitk_maxs);
        geti("VOLT_SPLITK_GROUPS", t.splitk_groups);
        geti("VOLT_SPLITK_MAXB", t.splitk_maxb);
        geti("VOLT_SCHED", t.sched);
        geti("VOLT_SCHED_MINB", t.sched_minb);
        geti("VOLT_SCHED_MAXB", t.sched_maxb);
        geti("VOLT_SCHED_MAXB_POTRF", t.sched_maxb_potrf);
        geti("VOLT_SCHED_G", t.sched_g);
        geti("VOLT_SCHED_S", t.sched_s);
        geti("VOLT_SCHED_GROUPS", t.sched_groups);
        geti("VOLT_SCHED_KMIN", t.sched_kmin);
        geti("VOLT_SMALL_NMAX", t.small_nmax);
        geti("VOLT_SMALL_MAXWG", t.small_maxwg);
        geti("VOLT_SMALL_PAD_MAXB", t.small_pad_maxb);
        geti("VOLT_SMALL_MAXB", t.small_maxb);
        geti("VOLT_SMALL_MAXB2", t.small_maxb2);
        geti("VOLT_LONG", t.long_on);
        geti("VOLT_LONG_FIRST", t.long_first);
        geti("VOLT_LONG_EMIN", t.long_emin);
        geti("VOLT_LONG_PAD", t.long_pad);
        geti("VOLT_LONG_NMIN", t.long_nmin);
        geti("VOLT_LONG_XCD", t.long_xcd);
        geti("VOLT_LONG_SPLIT", t.long_split);
        geti("VOLT_PLAIN_SPREAD", t.plain_spread);
        geti("VOLT_SPLIT_SPREAD", t.split_spread);
        geti("VOLT_BATCH", t.batch);
        geti("VOLT_BATCH_ORDER", t.batch_order);
        geti("VOLT_BATCH_LOCAL", t.batch_local);
        geti("VOLT_BATCH_SPREAD", t.batch_spread);
        geti("VOLT_BATCH_LAD", t.batch_lad);
        geti("VOLT_BATCH_PULLERS", t.batch_pullers);
        geti("VOLT_ROLLOUT_LANE", t.rollout_lane);
        geti("VOLT_LONG_PULLERS", t.long_pullers);
        geti("VOLT_BATCH_XSKEW", t.batch_xskew);
        geti("VOLT_BATCH_XDROP", t.batch_xdrop);
        geti("VOLT_BATCH64", t.batch64);
        geti("VOLT_BATCH64_MAX", t.batch64_max);
        geti("VOLT_BATCH64_MAX_STEP", t.batch64_max_step);
        if (const char* e = getenv("VOLT_SCHED_FRAC")) t.sched_frac = (float)atof(e);
        geti("VOLT_FAKE_CUS", t.cus);                        // tests: plan as if the device had this many CUs / XCDs
        geti("VOLT_FAKE_XCCS", t.xccs);
    return t;
}

// What the device looks like, and the gates that depend on it (host.h).  A process without a device (the CPU-side tests)
// keeps the full-chip defaults.
static void apply_topology(Tunables& t, bool faked) {
    if (!faked) {
        int dev = 0, cus = 0, xccs = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) {
            t.cus = cus;
            if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess && xccs > 0) t.xccs = xccs;
        } else {
            (void)hipGetLastError();
        }
    }
    if (t.cus == 256 && t.xccs == 8) return;
    const double f = t.cus / 256.0;
    auto scale = [f](int v, int lo) { const int r = (int)(v * f + 0.5); return r < lo ? lo : r; };
    t.sched_g = scale(t.sched_g, 8);
    t.plain_spread = scale(t.plain_spread, 8);
    t.split_spread = scale(t.split_spread, 16);
    t.splitk_target = scale(t.splitk_target, 16);
    t.group_gate = scale(t.group_gate, 16);
    t.batch_spread = scale(t.batch_spread, 8);
    t.small_nmax = 0;                                        // the one-launch steps: tuned for, and (batched step) placed on, the full chip
    t.long_on = 0;
    t.batch = 0;
    t.batch64 = 0;
}

const Tunables& tunables() {
    static const Tunables tn = [] {
        const char* tune0 = getenv("VOLT_TUNE");
        const bool tuning = tune0 && atoi(tune0) != 0;
        const bool faked = tuning && (getenv("VOLT_FAKE_CUS") || getenv("VOLT_FAKE_XCCS"));
        Tunables t = [&] {
            Tunables t;
            if (!tuning) return t;                           // the frozen defaults
            return read_env(t);
        }();
        apply_topology(t, faked);
        return t;
    }();
    return tn;
}
// Optional per-launch timing (bench only): every launch is bracketed by two events on ITS stream; the caller
// synchronises, and per kernel class gets the summed launch durations and the length of the UNION of the launch
// intervals (on one stream the two agree; with the batch cut into groups on several streams the launches of a class
// overlap, and the union is the time during which that class was running at all).
struct LaunchTimer {
    hipEvent_t base = nullptr;
    std::vector<hipEvent_t> ev;
    std::vector<int> cls;
    hipError_t err = hipSuccess;
    void note(hipError_t e) { if (err == hipSuccess && e != hipSuccess) err = e; }
    void start(hipStream_t s) {
        note(hipEventCreate(&base));
        note(hipEventRecord(base, s));
    }
    void begin(int c, hipStream_t s) {
        hipEvent_t e = nullptr;
        note(hipEventCreate(&e));
        note(hipEventRecord(e, s));
        ev.push_back(e);
        cls.push_back(c);
    }
    void end(hipStream_t s) {
        hipEvent_t e = nullptr;
        note(hipEventCreate(&e));
        note(hipEventRecord(e, s));
        ev.push_back(e);
    }
    // call after the device has drained
    void collect(float* ms_sum, float* ms_union, int* n_by_class, int nclass, float* per_launch = nullptr) {
        std::vector<std::vector<std::pair<float, float>>> iv(nclass);
        for (int c = 0; c < nclass; ++c) { ms_sum[c] = ms_union[c] = 0.f; n_by_class[c] = 0; }
        for (size_t i = 0; i < cls.size(); ++i) {
            float t0 = 0.f, t1 = 0.f;
            note(hipEventElapsedTime(&t0, base, ev[2 * i]));
            note(hipEventElapsedTime(&t1, base, ev[2 * i + 1]));
            ms_sum[cls[i]] += t1 - t0;
            n_by_class[cls[i]] += 1;
            if (per_launch) per_launch[i] = t1 - t0;
            iv[cls[i]].push_back({t0, t1});
        }
        for (int c = 0; c < nclass; ++c) {
            std::sort(iv[c].begin(), iv[c].end());
            float lo = 0.f, hi = -1.f;
            for (auto& x : iv[c]) {
                if (hi < lo || x.first > hi) {
                    if (hi >= lo) ms_union[c] += hi - lo;
                    lo = x.first;
                    hi = x.second;
                } else if (x.second > hi) {
                    hi = x.second;
                }
            }
            if (hi >= lo) ms_union[c] += hi - lo;
        }
        for (auto e : ev) (void)hipEventDestroy(e);
        if (base) (void)hipEventDestroy(base);
        ev.clear();
        cls.clear();
    }
};

// One factorisation (+ optional inverse).  Launch sequence: for every block column k ONE factor_step_kernel
//     [diagonal tile (k,k) | look-ahead for (k+1,k+1) | panel tiles (i,k): update + solve | trtri row k-1]
// and finally the trtri row n-1 alone.
struct FactorOpts {
    KSource src;            // src.K != nullptr: tiles of block columns >= 1 take their input straight from K
    float* Y;               // nullptr: no triangular inverse
    TriReduce red;          // red.rpad != nullptr: fuse z-partials and Frobenius partials into trtri
    SplitK sk;              // sk.slab != nullptr and sk.S > 1: small-batch launches cut their long products into K-slices
    const struct SchedDev* sched = nullptr;   // non-null: the host-balanced schedule (mid-size batches), needs sk.slab
};

// A balanced schedule: the items of all n (+1 with a triangular inverse) launches back to back, in PINNED HOST memory.
// The device copy lives in the CALLER's scratch (SplitK::tab), copied there once by volt_*_workspace_init -- the library
// owns no device memory.
constexpr int SCHED_HDR = 16;                  // int4 slots ahead of the items: the table's identity (two are used)
struct SchedDev {
    int4* items = nullptr;                   // pinned host: SCHED_HDR header slots, then the items
    size_t bytes = 0;                        // header + items
    int4 key[2];                             // {magic, B, n, inverse?}, {G, S, 1000 frac, min_len}: what the kernels check
    std::vector<int> item_off;               // per launch: where its items start (one more entry closes the last)
    int S = 0;
    int pad_lds = 0;                         // dynamic LDS bytes per workgroup: > 0 keeps it to one workgroup per CU
    int kmin = 0;                            // block columns below this run the plain one-tile-per-workgroup launch
};

// Built once per (device, B, n, inverse?, parameters) and kept for the life of the library, like the stream pool
// (host memory only).  A miss while the stream is being captured into a graph returns nullptr (no allocation inside a
// capture): the caller falls back to the schedules that need no tables.
static const SchedDev* get_sched(int B, int n, bool has_y, const SchedParams& p, hipStream_t s) {
    static std::mutex mu;
    static std::map<std::array<int, 8>, SchedDev*> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    const std::array<int, 8> key{dev, B, n, has_y ? 1 : 0, p.G, p.S, (int)(p.frac * 1000.f), p.min_len};
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    std::vector<SchedItem> items;
    SchedDev* sd = new SchedDev;
    sd->S = p.S;
    sd->pad_lds = 16 * 1024;                                 // 72 KB static + 16 KB: one workgroup per 160 KB CU
    // measured (N = 4096): below these block columns the plain launch is as fast or faster -- a cut tile pays ~25 us for
    // its slabs, and only the late launches (long trtri rows, k + 2 blocks against a mean of ~k / 2) are unbalanced enough
    // (ms/step at N = 4096 by first scheduled column, B = 3: 8 2.58, 12 2.57, 16 2.60; B = 6: 12 3.66, 16 3.64, 20 3.70;
    // B = 8: 16 4.03, 20 3.97; two groups of B / 2: 16)
    const int kmin_env = tunables().sched_kmin;
    // the factorisation alone (no trtri rows; per-call ms at N = 4096 by first scheduled column, B = 16: 12 3.95, 16 4.04,
    // 20 4.28, off 4.82; B = 32: 12 6.60, 20 6.64, off 7.93; B = 64: 12 12.48, 20 12.21, 24 12.22, off 12.87)
    sd->kmin = kmin_env >= 0 ? kmin_env : !has_y ? (B * (256 / p.G) >= 48 ? 20 : 12) : (p.G < 256 ? 16 : B <= 4 ? 12 : B <= 7 ? 16 : 20);
    const int launches = has_y ? n + 1 : n;
    for (int k = 0; k < launches; ++k) {
        sd->item_off.push_back((int)items.size());
        sched_build_launch(B, n, has_y, k, p, items);
    }
    sd->item_off.push_back((int)items.size());
    static_assert(sizeof(SchedItem) == sizeof(int4), "items are read as int4");
    sd->bytes = (SCHED_HDR + items.size()) * sizeof(SchedItem);
    sd->key[0] = int4{0x564f4c54, B, n, has_y ? 1 : 0};
    sd->key[1] = int4{p.G, p.S, (int)(p.frac * 1000.f), p.min_len};
    if (hipHostMalloc((void**)&sd->items, sd->bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        delete sd;
        sd = nullptr;
    } else {
        memset(sd->items, 0, SCHED_HDR * sizeof(SchedItem));
        sd->items[0] = sd->key[0];
        sd->items[1] = sd->key[1];
        memcpy(sd->items + SCHED_HDR, items.data(), items.size() * sizeof(SchedItem));
    }
    cache[key] = sd;                                         // a failure is remembered too: no retry per call
    return sd;
}

// Which balanced schedule B matrices of n block columns get (false: none).  Gs = stream groups it runs in, sp = the
// parameters of the per-group table; `cap` = slab rows the caller's scratch has.
static bool sched_choice(int B, int n, bool has_y, int cap, int& Gs, SchedParams& sp) {
    const Tunables& tn = tunables();
    if (!tn.sched || cap < 1 || B < tn.sched_minb || B > (has_y ? tn.sched_maxb : tn.sched_maxb_potrf)) return false;
    Gs = (tn.sched_groups > 1 && B >= 10 && B % tn.sched_groups == 0) ? tn.sched_groups : 1;
    sp = SchedParams();
    sp.G = tn.sched_g / Gs;
    sp.S = std::max(1, std::min(tn.sched_s, cap / B));       // the groups share the slab: cap / Gs rows for B / Gs matrices
    sp.frac = tn.sched_frac;
    (void)n;
    return true;
}

// The tables live in the CALLER's scratch, put there once by volt_*_workspace_init (below) -- the library owns no device
// memory, copies nothing per call and keeps NO record of which scratch it initialised (rounds 2-3 kept an address-keyed
// map: a region freed and handed out again at the same address was then taken for initialised).  The caller says so
// itself (the VOLT_WS_INITIALISED flag of the entry points that take a workspace); a factorisation is handed the table
// region only then, and every table-driven launch checks the header in the region against the table it expects, so scratch
// that was never initialised or was overwritten since is reported (info = INT_MIN + 1), never followed.
static int sched_install(void* tab, size_t tab_bytes, int B, int n, bool has_y, int cap, hipStream_t s) {
    int Gs = 1;
    SchedParams sp;
    if (!tab || !sched_choice(B, n, has_y, cap, Gs, sp)) return 0;
    const SchedDev* sd = get_sched(B / Gs, n, has_y, sp, s);
    if (!sd || sd->bytes > tab_bytes) return 0;
    hipError_t e = hipMemcpyAsync(tab, sd->items, sd->bytes, hipMemcpyHostToDevice, s);
    return e != hipSuccess ? (int)e : 0;
}

// Everything one group of matrices needs: the batch is cut into contiguous groups that run the same
// launch sequence on different streams (see run_factor_groups).
struct Group {
    float* A;
    float* Winv;
    int* info;
    FactorOpts o;
    int B;
    hipStream_t s;
};

// Timer classes: 0 = factor_step_kernel with a factorisation part (block columns 0..n-1, trtri row k-1 aboard),
//                1 = factor_step_kernel carrying only a trtri row (the last row; every row of volt_trtri_f32).
static void enqueue_step(const Group& g, int Np, int k, LaunchTimer* tm) {
    const int n = Np / TS, B = g.B;
    const TriReduce nored{nullptr, nullptr, nullptr, 0};
    const int itri = (g.o.Y && k > 0) ? k - 1 : -1;
    const int npre = (k >= 1 && k + 1 < n) ? B : 0;          // diagonal look-ahead workgroups
    const int grid = B + npre + (n - k - 1) * B + (itri >= 0 ? (itri + 1) * B : 0);
    if (g.o.sched && k >= g.o.sched->kmin) {                 // mid-size batch, late columns: the host's list, longest piece first
        const SchedDev& sd = *g.o.sched;
        SplitK sk = g.o.sk;
        sk.S = sd.S;
        sk.L = 1;
        for (int kk = k; kk <= (k + 1 == n && g.o.Y ? n : k); ++kk) {
            const int it = kk < n ? itri : n - 1;
            if (tm) tm->begin(kk < n ? 0 : 1, g.s);
            const int cnt = sd.item_off[kk + 1] - sd.item_off[kk];
            if (g.o.src.K && kk < n)
                hipLaunchKernelGGL(factor_step_sched_kernel<true>, dim3(cnt), dim3(256), sd.pad_lds, g.s, g.A, g.Winv, g.o.Y,
                                   g.info, Np, kk, it, B, g.o.src, g.o.Y ? g.o.red : nored, sk, g.o.sk.tab + SCHED_HDR + sd.item_off[kk], sd.key[0],
                                   sd.key[1]);
            else
                hipLaunchKernelGGL(factor_step_sched_kernel<false>, dim3(cnt), dim3(256), sd.pad_lds, g.s, g.A, g.Winv, g.o.Y,
                                   g.info, Np, kk, it, B, g.o.src, g.o.Y ? g.o.red : nored, sk, g.o.sk.tab + SCHED_HDR + sd.item_off[kk], sd.key[0],
                                   sd.key[1]);
            if (tm) tm->end(g.s);
        }
        return;
    }
    if (!g.o.sched && g.o.sk.slab && g.o.sk.S > 1) {         // small batch: every long product in K-slices
        // slices of about equal length: L blocks each, so that the launch has ~`target` of them (g.o.sk.S carries it)
        const int kk = k < n ? k : n - 1;
        const double blocks = (double)B * ((double)(n - kk - 1) * kk + 0.5 * kk * (kk - 1) + kk);   // panel + trtri + look-ahead
        int L = (int)(blocks / (double)g.o.sk.S + 0.999);
        const int minl = tunables().splitk_minl;
        if (L < minl) L = minl;
        int S = (kk + L - 1) / L;
        if (S < 1) S = 1;
        const int maxs = tunables().splitk_maxs;
        if (S > maxs) S = maxs;
        if (S * B > g.o.sk.cap) S = g.o.sk.cap / B;
        if (S < 1) S = 1;
        SplitK sk = g.o.sk;
        sk.S = S;
        sk.L = L;
        const int gs = B + S * (npre + (n - k - 1) * B + (itri >= 0 ? (itri + 1) * B : 0));
        if (tm) tm->begin(0, g.s);
        // (as the plain launches below: up to 700 slices one workgroup per CU -- two series of N = 4096 2.30 -> 2.12 ms/step)
        const int split_spread = tunables().split_spread;
        const unsigned spad = gs <= split_spread ? 16 * 1024 : 0;
        if (g.o.src.K)
            hipLaunchKernelGGL(factor_step_split_kernel<true>, dim3(gs), dim3(256), spad, g.s, g.A, g.Winv, g.o.Y, g.info, Np,
                               k, itri, B, g.o.src, g.o.Y ? g.o.red : nored, sk);
        else
            hipLaunchKernelGGL(factor_step_split_kernel<false>, dim3(gs), dim3(256), spad, g.s, g.A, g.Winv, g.o.Y, g.info, Np,
                               k, itri, B, g.o.src, g.o.Y ? g.o.red : nored, sk);
        if (tm) tm->end(g.s);
        if (k + 1 == n && g.o.Y) {                           // the trailing trtri row: k = n (no factorisation part)
            if (tm) tm->begin(1, g.s);
            hipLaunchKernelGGL(factor_step_split_kernel<false>, dim3(S * n * B), dim3(256), S * n * B <= split_spread ? 16 * 1024 : 0, g.s, g.A, g.Winv, g.o.Y,
                               g.info, Np, n, n - 1, B, g.o.src, g.o.red, sk);
            if (tm) tm->end(g.s);
        }
        return;
    }
    if (tm) tm->begin(0, g.s);
    // A launch of about one tile per CU is spread out (16 KB of LDS padding: one workgroup per CU): left alone the dispatcher
    // pairs tiles up on some CUs, where two share one MFMA pipe, and the launch lasts as long as the slower pairs.  ms/step at
    // N = 4096 without -> with: B = 9 5.51 -> 5.09, 10 4.90 -> 4.51, 12 5.45 -> 5.14, 14 6.09 -> 5.87, 16 6.59 -> 6.51, 20 (330
    // tiles per launch) 8.01 -> 8.04, 24 (396) 9.19 -> 9.39, 32 (528) 11.6 -> 12.3; B <= 8 unchanged: up to 320 tiles
    const unsigned pad = grid <= tunables().plain_spread ? 16 * 1024 : 0;
    if (g.o.src.K)
        hipLaunchKernelGGL(factor_step_kernel<true>, dim3(grid), dim3(256), pad, g.s, g.A, g.Winv, g.o.Y, g.info, Np, k,
                           itri, B, g.o.src, g.o.Y ? g.o.red : nored);
    else
        hipLaunchKernelGGL(factor_step_kernel<false>, dim3(grid), dim3(256), pad, g.s, g.A, g.Winv, g.o.Y, g.info, Np, k,
                           itri, B, g.o.src, g.o.Y ? g.o.red : nored);
    if (tm) tm->end(g.s);
    if (k + 1 == n && g.o.Y) {
        if (tm) tm->begin(1, g.s);
        hipLaunchKernelGGL(factor_step_kernel<false>, dim3(n * B), dim3(256), 0, g.s, g.A, g.Winv, g.o.Y, g.info, Np,
                           -1, n - 1, B, g.o.src, g.o.red);
        if (tm) tm->end(g.s);
    }
}

// info = 0 and every W block's ready flag cleared, on the caller's stream before anything forks from it
static int begin_factor(float* Winv, int* info, int B, int n, hipStream_t s, int* sk_count = nullptr) {
    const int nflags = B * n, ncount = sk_count ? (n + 1) * (n + 1) * B : 0;   // counters: [n+1 launches][B (n+1) tiles]
    int blocks = (std::max(std::max(nflags, B), ncount) + 255) / 256;
    if (blocks > 256) blocks = 256;
    if (blocks * 256 < std::max(nflags, B)) blocks = (std::max(nflags, B) + 255) / 256;
    hipLaunchKernelGGL(begin_factor_kernel, dim3(blocks), dim3(256), 0, s, Winv, nflags, info, B, sk_count, ncount);
    return 0;
}

// ---- stage barriers vs. asynchrony ------------------------------------------------------------------
// With the whole batch in lockstep every launch ends in a tail: its tiles rarely fill a whole number of
// rounds of the 512 resident workgroups, and the next stage cannot start before the tail has drained.
// Cutting the batch into G groups that run the SAME launch sequence on G streams lets one group's tail
// overlap another group's next stage.  The auxiliary streams and the fork / join events are created once per
// device (library-lifetime, like a BLAS handle's); a call forks from and joins back into the caller's stream
// with those events, so the caller still sees ordinary stream semantics.  Calls on one device are serialised
// on the HOST while they enqueue (a mutex around fork .. join: the events are shared), never on the device.
constexpr int MAX_GROUPS = 8;
struct StreamPool {
    hipStream_t aux[MAX_GROUPS - 1];
    hipEvent_t fork, join[MAX_GROUPS - 1], extra[5];     // extra: chol64.hip's look-ahead schedules
    std::mutex mu;
    int want_groups = 2;                 // tunables().groups
    bool ok = false;
};
static StreamPool* stream_pool() {
    static StreamPool pools[16];
    static std::once_flag once[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::call_once(once[dev], [dev]() {
        StreamPool& p = pools[dev];
        bool ok = hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < MAX_GROUPS - 1; ++i) {
            ok = ok && hipStreamCreateWithFlags(&p.aux[i], hipStreamNonBlocking) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&p.join[i], hipEventDisableTiming) == hipSuccess;
        }
        for (int i = 0; i < 5; ++i) ok = ok && hipEventCreateWithFlags(&p.extra[i], hipEventDisableTiming) == hipSuccess;
        p.want_groups = tunables().groups;
        if (p.want_groups < 1) p.want_groups = 1;
        if (p.want_groups > MAX_GROUPS) p.want_groups = MAX_GROUPS;
        p.ok = ok;
    });
    return pools[dev].ok ? &pools[dev] : nullptr;
}

// chol64.hip runs its look-ahead on the same pool: one auxiliary stream, the fork event and five of the join events,
// under the pool's mutex while it enqueues -- like run_factor_groups below.
struct VoltAux {
    hipStream_t aux, aux2, aux3, aux4;
    hipEvent_t fork, ev[12];
    std::mutex* mu;
};
bool volt_internal_aux(VoltAux* out) {
    StreamPool* p = stream_pool();
    if (!p) return false;
    // (lowest-priority streams for the bulk work were tried: the chain then waits for events from a stream the hardware
    // serves last -- one 4096^2 matrix 4.2 -> 8.1 ms)
    *out = VoltAux{p->aux[0], p->aux[1], p->aux[2], p->aux[3], p->fork,
                   {p->join[0], p->join[1], p->join[2], p->join[3], p->join[4], p->join[5], p->join[6], p->extra[0],
                    p->extra[1], p->extra[2], p->extra[3], p->extra[4]}, &p->mu};
    return true;
}

static int pick_groups(const StreamPool* pool, int B, int force) {
    int want = force > 0 ? force : (pool ? pool->want_groups : 1);
    if (want > MAX_GROUPS) want = MAX_GROUPS;
    while (want > 1 && (B % want != 0 || B / want < 8)) want >>= 1;   // keep whole-XCD groups of >= 8 matrices
    return want;
}

typedef void (*volt_group_post_fn)(void* ctx, int b0, int Bg, hipStream_t s);

#define VOLT_TRY(call)                              \
    do {                                            \
        hipError_t e__ = (call);                    \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

static int run_factor_groups(float* A, float* Winv, int* info, int B, int Np, hipStream_t s, const FactorOpts& o,
                             volt_group_post_fn post = nullptr, void* post_ctx = nullptr, LaunchTimer* tm = nullptr,
                             int force_groups = 0) {
    const int n = Np / TS;
    StreamPool* pool = stream_pool();
    int G = pool ? pick_groups(pool, B, force_groups) : 1;
    // Short series: a launch of B (n + 1) tiles that does not fill the 512 workgroup slots gains nothing from sharing the
    // chip with a second group and pays its launches twice (64 x N=399: 0.289 ms/step as one group, 0.387 as two; 64 x
    // 1000: 0.884 / 0.926; 64 x 1400: 1.85 / 1.62; 512 x 399: 1.12 / 1.09)
    if (force_groups == 0 && (int64_t)B * (n + 1) < tunables().group_gate) G = 1;
    // Inside a graph capture the groups become branches of the graph, and how the runtime maps them back onto streams at
    // replay is not ours to say: 64 x 4096 replayed at 22.5 ms in some processes and at 34.8 ms in others.  One group is
    // predictable (25.5 ms); captured loops are for the launch-bound sizes, which run as one group anyway.
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cap_status) == hipSuccess && cap_status != hipStreamCaptureStatusNone;
    if (capturing && force_groups == 0) G = 1;
    // Small batches: cut the long products into K-slices so that a launch has ~`target` workgroups (measured, N = 4096,
    // ms/step unsplit -> split: B = 1 4.5 -> 2.0, 2 4.6 -> 2.6, 4 4.7 -> 3.4, 6 4.7 -> 4.4; from B = 8 on a launch has a
    // tile per CU and splitting on ONE stream stops paying: B = 8 stays unsplit, 4.8 ms).  10 <= B <= 20: two split groups
    // on two streams up to B = 20 (B = 12: 8.4 -> 6.1 ms, 16: 8.4 -> 7.1, 20: 9.6 -> 9.1; no gain from 24 on).
    const Tunables& tn = tunables();
    const int target = tn.splitk_target, split_groups = tn.splitk_groups, split_maxb = tn.splitk_maxb;
    const bool can_split = o.sk.slab && o.sk.count && force_groups == 0 && target > 1;
    FactorOpts o1 = o;
    o1.sk.S = 1;                                             // > 1: the split schedule, and the slices wanted per launch
    if (can_split && B < 8) {
        G = 1;
        o1.sk.S = target;
    }
    if (can_split && !capturing && B >= 10 && B < split_maxb && pool && split_groups > 1 && B % split_groups == 0) {
        G = split_groups;
        o1.sk.S = target / G;
    }
    // 3 <= B < 32: the late block columns run the host-balanced schedule (sched.h), one group up to B = 9, two groups on
    // two streams from 10 on; the early columns the plain launch.  Measured, N = 4096, ms/step before -> after: B = 3
    // 2.90 -> 2.57, 4 3.24 -> 3.00, 6 4.21 -> 3.64, 7 4.76 -> 3.83, 8 4.60 -> 3.97, 12 5.90 -> 5.43, 20 8.79 -> 8.02,
    // 24 9.99 -> 9.20, 28 11.08 -> 10.46 (16: no change).  B = 1, 2 stay on the all-split schedule below.
    if (can_split) {
        int Gs = 1;
        SchedParams sp;
        // o.sk.tab != nullptr: the caller declared its scratch initialised -- the table volt_*_workspace_init copied there
        // is the one get_sched returns for this shape (a pure function of it; the launches check the header on the device)
        const SchedDev* sd = nullptr;
        if (o.sk.tab && sched_choice(B, n, o.Y != nullptr, o.sk.cap, Gs, sp) && (Gs == 1 || (pool && !capturing)))
            sd = get_sched(B / Gs, n, o.Y != nullptr, sp, s);
        // short matrices never reach the scheduled columns; below 8 matrices the alternative is the all-split schedule,
        // which is the better one while most columns are early ones (B = 4, n = 16: 0.92 ms all-split, 1.04 hybrid)
        if (sd && sd->bytes <= o.sk.tab_bytes && n > sd->kmin + (B < 8 ? 7 : 1)) {
            // (the early columns as all-split launches instead of plain ones were measured too: no better, B = 7 4.13 vs 3.82)
            G = Gs;
            o1.sk.S = 2;                                     // > 1: the counters are cleared below, the slab is shared out
            o1.sched = sd;
        }
    }
    int rc = begin_factor(Winv, info, B, n, s, o1.sk.S > 1 ? o1.sk.count : nullptr);
    if (rc) return rc;
    std::unique_lock<std::mutex> lock;
    if (pool && G > 1) lock = std::unique_lock<std::mutex>(pool->mu);
    if (pool && G > 1) VOLT_TRY(hipEventRecord(pool->fork, s));
    if (tm) tm->start(s);
    if (G == 1) {
        const Group g{A, Winv, info, o1, B, s};
        for (int k = 0; k < n; ++k) enqueue_step(g, Np, k, tm);
        if (post) post(post_ctx, 0, B, s);
        VOLT_LAUNCH_CHECK();
        return tm && tm->err != hipSuccess ? (int)tm->err : 0;
    }
    const int Bg = B / G;
    const int64_t mat = (int64_t)Np * Np;
    Group grp[MAX_GROUPS];
    for (int g = 0; g < G; ++g) {
        FactorOpts og = o1;
        const int b0 = g * Bg;
        if (og.sk.S > 1) {                                   // each group its own share of the slab and of the counters
            og.sk.slab += (int64_t)g * (o.sk.cap / G) * (n + 1) * TS * TS;
            og.sk.count += (int64_t)g * (n + 1) * (n + 1) * Bg;
            og.sk.cap = o.sk.cap / G;
        }
        if (og.src.K) {
            og.src.K += (int64_t)b0 * og.src.bsk;
            if (og.src.sigma2) og.src.sigma2 += b0;
        }
        if (og.Y) og.Y += b0 * mat;
        if (og.red.rpad) {
            og.red.rpad += (int64_t)b0 * Np;
            og.red.zpart += (int64_t)b0 * n * Np;
            og.red.frob += (int64_t)b0 * (n * (n + 1) / 2);
        }
        grp[g] = Group{A + b0 * mat, Winv + (int64_t)b0 * n * TS * TS, info + b0, og, Bg, g == 0 ? s : pool->aux[g - 1]};
        if (g > 0) VOLT_TRY(hipStreamWaitEvent(grp[g].s, pool->fork, 0));
    }
    for (int k = 0; k < n; ++k)
        for (int g = 0; g < G; ++g) enqueue_step(grp[g], Np, k, tm);
    if (post)
        for (int g = 0; g < G; ++g) post(post_ctx, g * Bg, Bg, grp[g].s);
    // join: the caller's stream must not run ahead of any group.  If an event call fails the group is joined on the
    // host instead, so the caller's stream semantics hold either way; the error is still reported.
    int first_err = 0;
    for (int g = 1; g < G; ++g) {
        hipError_t e = hipEventRecord(pool->join[g - 1], grp[g].s);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, pool->join[g - 1], 0);
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(grp[g].s);
            if (!first_err) first_err = (int)e;
        }
    }
    if (first_err) return first_err;
    VOLT_LAUNCH_CHECK();
    return tm && tm->err != hipSuccess ? (int)tm->err : 0;
}

static int run_trtri(const float* A, const float* Winv, float* Y, int B, int Np, hipStream_t s, LaunchTimer* tm) {
    const int n = Np / TS;
    const TriReduce nored{nullptr, nullptr, nullptr, 0};
    for (int i = 0; i < n; ++i) {
        if (tm) tm->begin(1, s);
        hipLaunchKernelGGL(factor_step_kernel<false>, dim3((i + 1) * B), dim3(256), 0, s, const_cast<float*>(A),
                           const_cast<float*>(Winv), Y, nullptr, Np, -1, i, B, KSource{nullptr, 0, 0, nullptr, 0.f, 0},
                           nored);
        if (tm) tm->end(s);
    }
    VOLT_LAUNCH_CHECK();
    return 0;
}

// Upper bound of the balanced schedule's tables for B matrices of n block columns (0 where it never runs): n + 1 launches
// of at most B diagonal items + 4 slices of B (n + 1) tiles.
size_t volt_internal_sched_bytes(int B, int n) {
    if (B < 3 || B > 64 || n < 8) return 0;
    return ((((size_t)(n + 1) * B * (1 + 4 * (size_t)(n + 1)) + SCHED_HDR) * sizeof(SchedItem)) + 255) & ~(size_t)255;
}

// used by mll.hip (volt_mll_workspace_init_f32)
int volt_internal_sched_install(void* tab, size_t tab_bytes, int B, int n, int has_y, int cap, void* stream) {
    return sched_install(tab, tab_bytes, B, n, has_y != 0, cap, (hipStream_t)stream);
}

// used by mll.hip
int volt_internal_factor(const float* K, int64_t ldk, int64_t bsk, const float* sigma2, float jitter, float* A,
                         float* Winv, float* Y, int* info, const float* rpad, float* zpart, float* frob, int B, int N,
                         void* stream, volt_group_post_fn post, void* post_ctx, float* sk_slab, int* sk_count, int sk_rows,
                         void* tab, size_t tab_bytes) {
    const int Np = volt_padded_n(N), n = Np / TS;
    hipStream_t s = (hipStream_t)stream;
    // block column 0 (its diagonal tile is factored straight out of A) is copied; everything else is read from K
    hipLaunchKernelGGL(prepare_kernel, dim3(n, B), dim3(256), 0, s, K, ldk, bsk, sigma2, jitter, A, N, Np, 1);
    FactorOpts o{KSource{K, ldk, bsk, sigma2, jitter, N}, Y, TriReduce{rpad, zpart, frob, N},
                 SplitK{sk_slab, sk_count, 1, 1, sk_rows, (int4*)tab, tab_bytes}};
    return run_factor_groups(A, Winv, info, B, Np, s, o, post, post_ctx);
}

// The same, with every launch bracketed by HIP events on its own stream (bench.py's roofline leg through
// volt_profile_step_f32 in mll.hip): identical buffers, reductions and scratch, so the profiled launches ARE the
// timed step's -- including, through `post`, the O(N^2) tail each group runs on its own stream beside the other groups'
// factor launches (it is not timed itself, but it shares the GPU with the launches that are).  Synchronises the stream.
int volt_internal_profile(const float* K, int64_t ldk, int64_t bsk, const float* sigma2, float* A, float* Winv, float* Y,
                          int* info, const float* rpad, float* zpart, float* frob, int B, int N, int groups, void* stream,
                          float* sk_slab, int* sk_count, int sk_rows, void* tab, size_t tab_bytes, volt_group_post_fn post,
                          void* post_ctx, float* ms_sum_host, float* ms_union_host, int* launches_host,
                          float* per_launch_host) {
    if (groups < 0 || groups > MAX_GROUPS) return -11;
    const int Np = volt_padded_n(N), n = Np / TS;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(prepare_kernel, dim3(n, B), dim3(256), 0, s, K, ldk, bsk, sigma2, 0.f, A, N, Np, 1);
    FactorOpts o{KSource{K, ldk, bsk, sigma2, 0.f, N}, Y, TriReduce{rpad, zpart, frob, N},
                 SplitK{sk_slab, sk_count, 1, 1, sk_rows, (int4*)tab, tab_bytes}};
    LaunchTimer tm;
    const int rc = run_factor_groups(A, Winv, info, B, Np, s, o, post, post_ctx, &tm, groups);
    hipError_t e = hipStreamSynchronize(s);            // the groups have joined into s
    tm.collect(ms_sum_host, ms_union_host, launches_host, 2, per_launch_host);
    if (rc) return rc;
    if (e != hipSuccess) return (int)e;
    return tm.err != hipSuccess ? (int)tm.err : 0;
}

extern "C" {

int volt_prepare_f32(const float* K, int64_t ldk, int64_t bsk, const float* sigma2, float jitter, float* A, int B,
                     int N, void* stream) {
    if (!K) return -1;
    if (ldk < N) return -2;
    if (!A) return -6;
    if (B < 0 || B > 65535) return -7;          // the batch rides in gridDim.y
    if (N < 1) return -8;
    if (B == 0) return 0;
    const int Np = volt_padded_n(N), n = Np / TS;
    hipLaunchKernelGGL(prepare_kernel, dim3(n * (n + 1) / 2, B), dim3(256), 0, (hipStream_t)stream, K, ldk, bsk,
                       sigma2, jitter, A, N, Np, 0);
    VOLT_LAUNCH_CHECK();
    return 0;
}

int volt_tune_update_f32(float* A, const float* Winv, int* info, int B, int Np, int k, int var, int reps, void* stream) {
    if (!A) return -1;
    if (!Winv) return -2;
    if (!info) return -3;
    if (B < 1) return -4;
    if (Np < TS || Np % TS) return -5;
    const int n = Np / TS;
    if (k < 1 || k + 1 >= n) return -6;
    if (var < 0 || var > 5) return -7;
    hipStream_t s = (hipStream_t)stream;
    const KSource none{nullptr, 0, 0, nullptr, 0.f, 0};
    for (int r = 0; r < reps; ++r) {
        if (var == 0) hipLaunchKernelGGL(tune_update_kernel, dim3((n - k - 1) * B), dim3(256), 0, s, A, Winv, info, Np, k, B, none);
        else if (var == 1) hipLaunchKernelGGL(tune_update_sq_kernel<0>, dim3((n - k - 1) * B), dim3(256), 0, s, A, Np, k, B, none);
        else if (var == 2) hipLaunchKernelGGL(tune_update_sq_kernel<1>, dim3((n - k - 1) * B), dim3(256), 0, s, A, Np, k, B, none);
        else if (var == 3) hipLaunchKernelGGL(tune_update_sq_kernel<2>, dim3((n - k - 1) * B), dim3(256), 0, s, A, Np, k, B, none);
        else if (var == 4) hipLaunchKernelGGL(tune_update_sq_kernel<3>, dim3((n - k - 1) * B), dim3(256), 0, s, A, Np, k, B, none);
        else hipLaunchKernelGGL(tune_empty_kernel, dim3((n - k - 1) * B), dim3(256), 0, s, A);
    }
    VOLT_LAUNCH_CHECK();
    return 0;
}

int volt_sched_describe(int B, int n, int has_y, int k, int G, int S, float frac, int* items, int max_items, float* loads) {
    if (B < 1 || n < 1 || k < 0 || k > n || (k == n && !has_y) || G < 1 || S < 1) return -1;
    SchedParams p;
    p.G = G;
    p.S = S;
    if (frac > 0.f) p.frac = frac;
    std::vector<SchedItem> it;
    std::vector<float> ld;
    sched_build_launch(B, n, has_y != 0, k, p, it, &ld);
    if ((int)it.size() > max_items) return -2;
    if (items) memcpy(items, it.data(), it.size() * sizeof(SchedItem));
    if (loads) memcpy(loads, ld.data(), ld.size() * sizeof(float));
    return (int)it.size();
}



int volt_tune_diag_f32(float* A, float* Winv, int* info, int B, int Np, int k, long long* stamps, void* stream) {
    if (!A) return -1;
    if (!Winv) return -2;
    if (!info) return -3;
    if (B < 1) return -4;
    if (Np < TS || Np % TS) return -5;
    if (k < 0 || k >= Np / TS) return -6;
    if (!stamps) return -7;
    hipLaunchKernelGGL(tune_diag_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, A, Winv, info, Np, k, stamps);
    VOLT_LAUNCH_CHECK();
    return 0;
}

// Scratch of the factorisation alone: slab rows (of n + 1 tiles each) and arrival counters.  Without the trtri rows a
// launch has only B (n - k + 1) tiles, so even 64 matrices leave CUs idle in the late block columns: up to 31 matrices
// get the 64 rows the MLL step's workspace has, 32..64 get 128 (two slices per tile for 64 matrices).
static int potrf_ws_rows(int B) { return B < 32 ? VOLT_SPLITK_SLABS : (B <= 64 ? 2 * VOLT_SPLITK_SLABS : 0); }
static size_t potrf_ws_slab_bytes(int B, int Np) {
    return (((size_t)potrf_ws_rows(B) * (Np / TS + 1) * TS * TS * sizeof(float)) + 255) & ~(size_t)255;
}

static size_t potrf_ws_count_bytes(int B, int Np) {
    const size_t n = (size_t)Np / TS;
    return (((n + 1) * (n + 1) * B * sizeof(int)) + 255) & ~(size_t)255;
}
// scratch of the launch-per-column schedules (slabs, counters, balanced tables) ...
static size_t potrf_ws_sched_bytes(int B, int Np) {
    if (B < 1 || potrf_ws_rows(B) == 0 || Np < TS || Np % TS) return 0;   // more than 64 matrices fill the chip with whole tiles
    if (Np / TS < 3) return 0;               // k <= 1: no product is long enough to be cut (slices are >= 2 K-blocks)
    return potrf_ws_slab_bytes(B, Np) + potrf_ws_count_bytes(B, Np) + volt_internal_sched_bytes(B, Np / TS);
}
// ... followed by the table + progress words of the one-launch factorisation (volt_potrf_k_f32 only: tiles read from K)
size_t volt_potrf_workspace_bytes(int B, int Np) {
    if (B < 1 || Np < TS || Np % TS) return 0;
    return potrf_ws_sched_bytes(B, Np) + volt_internal_batch_bytes(B, Np / TS, 0);
}

int volt_potrf_workspace_init_f32(void* ws, size_t ws_bytes, int B, int Np, void* stream) {
    if (B < 1) return -3;
    if (Np < TS || Np % TS) return -4;
    const size_t need = volt_potrf_workspace_bytes(B, Np);
    if (!need) return 0;
    if (!ws || ((uintptr_t)ws & 255)) return -1;
    if (ws_bytes < need) return -2;
    const size_t bb = volt_internal_batch_bytes(B, Np / TS, 0);
    if (bb) {
        const int rc = volt_internal_batch_install(reinterpret_cast<char*>(ws) + potrf_ws_sched_bytes(B, Np), bb, B, Np / TS, 0, stream);
        if (rc) return rc;
    }
    const size_t tb = potrf_ws_sched_bytes(B, Np) ? volt_internal_sched_bytes(B, Np / TS) : 0;
    if (!tb) return 0;
    return sched_install(reinterpret_cast<char*>(ws) + potrf_ws_slab_bytes(B, Np) + potrf_ws_count_bytes(B, Np), tb, B, Np / TS,
                         false, potrf_ws_rows(B), (hipStream_t)stream);
}

int volt_potrf_ws_f32(float* A, float* Winv, int* info, int B, int Np, void* ws, size_t ws_bytes, int ws_flags, void* stream) {
    if (!A) return -1;
    if (!Winv) return -2;
    if (!info) return -3;
    if (B < 0) return -4;
    if (Np < TS || Np % TS) return -5;
    if (B == 0) return 0;
    SplitK sk{nullptr, nullptr, 1, 1, 0, nullptr, 0};
    const size_t need = volt_potrf_workspace_bytes(B, Np);
    if (ws) {
        if (((uintptr_t)ws & 255) != 0) return -6;
        if (ws_bytes < need) return -7;
        // the whole factorisation in one launch (batch_step.hip: the tiles read their input from A itself), on the caller's word
        // that the init ran on this scratch -- the same schedule, and so the same bits, as volt_potrf_k_f32 on the same matrix
        const size_t bb = volt_internal_batch_bytes(B, Np / TS, 0);
        if (bb && (ws_flags & VOLT_WS_INITIALISED)) {
            const int rc = volt_internal_batch_step(nullptr, 0, 0, nullptr, 0.f, A, Winv, nullptr, info, nullptr, nullptr, nullptr, B, Np,
                                                    nullptr, nullptr, reinterpret_cast<char*>(ws) + potrf_ws_sched_bytes(B, Np), bb,
                                                    stream, nullptr, nullptr);
            if (rc == 1) return 0;
            if (rc) return rc > 0 ? rc : -8;
        }
        if (potrf_ws_sched_bytes(B, Np)) {                       // K-slices and the balanced schedule
            sk.slab = reinterpret_cast<float*>(ws);
            sk.count = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + potrf_ws_slab_bytes(B, Np));
            sk.cap = potrf_ws_rows(B);
            sk.tab_bytes = volt_internal_sched_bytes(B, Np / TS);
            // the table region is followed only on the caller's word that volt_potrf_workspace_init_f32 ran on this scratch
            sk.tab = (sk.tab_bytes && (ws_flags & VOLT_WS_INITIALISED))
                         ? reinterpret_cast<int4*>(reinterpret_cast<char*>(ws) + potrf_ws_slab_bytes(B, Np) + potrf_ws_count_bytes(B, Np))
                         : nullptr;
        }
    }
    FactorOpts o{KSource{nullptr, 0, 0, nullptr, 0.f, 0}, nullptr, TriReduce{nullptr, nullptr, nullptr, 0}, sk};
    return run_factor_groups(A, Winv, info, B, Np, (hipStream_t)stream, o);
}

int volt_potrf_f32(float* A, float* Winv, int* info, int B, int Np, void* stream) {
    return volt_potrf_ws_f32(A, Winv, info, B, Np, nullptr, 0, 0, stream);
}

// Factor of K + (sigma2 + jitter) I straight from K: only block column 0 is copied, every other tile is read from K by
// the workgroup that updates it (as the MLL step does) -- volt_prepare_f32's pass over the lower triangle (2.2 GB in,
// 2.2 GB out and 0.82 ms for 64 x 4096^2: 7 % of the factorisation) disappears.
int volt_potrf_k_f32(const float* K, int64_t ldk, int64_t bsk, const float* sigma2, float jitter, float* A, float* Winv,
                     int* info, int B, int N, void* ws, size_t ws_bytes, int ws_flags, void* stream) {
    if (!K) return -1;
    if (ldk < N) return -2;
    if (!A) return -6;
    if (!Winv) return -7;
    if (!info) return -8;
    if (B < 0 || B > 65535) return -9;
    if (N < 1) return -10;
    if (B == 0) return 0;
    const int Np = volt_padded_n(N), n = Np / TS;
    SplitK sk{nullptr, nullptr, 1, 1, 0, nullptr, 0};
    const size_t need = volt_potrf_workspace_bytes(B, Np);
    if (ws) {
        if (((uintptr_t)ws & 255) != 0) return -11;
        if (ws_bytes < need) return -12;
        // the whole factorisation in one launch (batch_step.hip), on the caller's word that the init ran on this scratch
        const size_t bb = volt_internal_batch_bytes(B, n, 0);
        if (bb && (ws_flags & VOLT_WS_INITIALISED)) {
            const int rc = volt_internal_batch_step(K, ldk, bsk, sigma2, jitter, A, Winv, nullptr, info, nullptr, nullptr, nullptr, B, N,
                                                    nullptr, nullptr, reinterpret_cast<char*>(ws) + potrf_ws_sched_bytes(B, Np), bb,
                                                    stream, nullptr, nullptr);
            if (rc == 1) return 0;
            if (rc) return rc > 0 ? rc : -13;
        }
        if (potrf_ws_sched_bytes(B, Np)) {
            sk.slab = reinterpret_cast<float*>(ws);
            sk.count = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + potrf_ws_slab_bytes(B, Np));
            sk.cap = potrf_ws_rows(B);
            sk.tab_bytes = volt_internal_sched_bytes(B, Np / TS);
            // the table region is followed only on the caller's word that volt_potrf_workspace_init_f32 ran on this scratch
            sk.tab = (sk.tab_bytes && (ws_flags & VOLT_WS_INITIALISED))
                         ? reinterpret_cast<int4*>(reinterpret_cast<char*>(ws) + potrf_ws_slab_bytes(B, Np) + potrf_ws_count_bytes(B, Np))
                         : nullptr;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(prepare_kernel, dim3(n, B), dim3(256), 0, s, K, ldk, bsk, sigma2, jitter, A, N, Np, 1);
    FactorOpts o{KSource{K, ldk, bsk, sigma2, jitter, N}, nullptr, TriReduce{nullptr, nullptr, nullptr, 0}, sk};
    return run_factor_groups(A, Winv, info, B, Np, s, o);
}

int volt_trtri_f32(const float* A, const float* Winv, float* Y, int B, int Np, void* stream) {
    if (!A) return -1;
    if (!Winv) return -2;
    if (!Y) return -3;
    if (B < 0) return -4;
    if (Np < TS || Np % TS) return -5;
    if (B == 0) return 0;
    return run_trtri(A, Winv, Y, B, Np, (hipStream_t)stream, nullptr);
}

}  // extern "C"
