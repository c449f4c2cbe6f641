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

// Slabs are stored WRITE-THROUGH (sc1): the data goes to memory without a release fence.  A release
// (buffer_wbl2) writes back every dirty line of the XCD's L2 -- with hundreds of slices arriving per launch, each
// behind its own release, the split schedule ran SLOWER the more slices there were (B = 8: S = 2 5.1 ms, S = 8 9.0 ms).
__device__ __forceinline__ void slab_dump(const f32x16 (&acc)[4], float* __restrict__ slab) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, TS * TS * 4, 0x00020000);
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[t4][4 * g + e];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, ((t4 * 4 + g) * NT + (int)threadIdx.x) * 16, 0,
                                                   16 /* sc1 */);
        }
}
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
        float a = 0.f;
        for (int jb = 0; jb <= i; ++jb) a += red.zpart[((int64_t)b * n + jb) * Np + i * TS + tid];
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
    for (int c = tid; c < Np; c += NT) {
        float al = 0.f;
        for (int i = c / TS; i < n; ++i) al += tl.apart[((int64_t)b * n + i) * Np + c];
        tl.apad[(int64_t)b * Np + c] = al;
        if (c < N) {
            const double zi = tl.z[(int64_t)b * Np + c];
            v[0] += zi * zi;
            v[1] += log((double)Ab[(int64_t)c * Np + c]);
            v[2] += (double)al * al;
            tl.alpha[(int64_t)b * N + c] = al;
        }
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
                                                           KSource src, TriReduce red, SmallState st, SmallTail tl) {
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
    int w = blockIdx.x, b = 0, k = 0, i = 0, j = 0, kind = -1;          // kind 0: D(0) / S(k)  1: P(i,k)  2: T(i,j)  3: U(k)
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

__global__ __launch_bounds__(256, 1) void long_step_kernel(float* __restrict__ A, float* __restrict__ Winv,
                                                          float* __restrict__ Y, int* __restrict__ info, int Np,
                                                          KSource src, TriReduce red, LongState st, SmallTail tl) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    __shared__ int s_last;
    const int n = Np / TS, tid = threadIdx.x;
    if (st.hdr[0] != SMALL_MAGIC || st.hdr[1] != 1 || st.hdr[2] != n) {          // not (or no longer) what init wrote
        if (tid == 0 && blockIdx.x == 0) info[0] = (int)0x80000001;
        return;
    }
    const int want = __hip_atomic_load(st.hdr + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    const int4 item = st.items[blockIdx.x];
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
    SMALL_STAMP(0);
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
        SMALL_STAMP(1);
        const float* Lt = Ab + (int64_t)k * TS * Np + (int64_t)(k - 1) * TS;
        const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)Lt, 0, 0x7fffffff, 0x00020000);
        float* sL = smem;
        const int row = tid >> 1, half = tid & 1;
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            if (tid == 0 && !wait_flag(hf + 4 * k + j, 4 * want, 2)) atomicCAS(info_b, 0, (int)0x80000000);
            __syncthreads();
            if (st.stamps && tid == 0) st.stamps[(int64_t)blockIdx.x * 16 + 6 + j] = __builtin_amdgcn_s_memrealtime();
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
        SMALL_STAMP(3);
        diag_body<false, true>(A, Winv, info, Np, k, 0, smem, nullptr, true, wf + k, want, sf + 4 * k);
        SMALL_STAMP(4);
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
        SMALL_STAMP(2);
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
                                        st.stamps ? st.stamps + (int64_t)blockIdx.x * 16 : nullptr, NoOp(), hf + 4 * pa);
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
                                                     st.stamps ? st.stamps + (int64_t)blockIdx.x * 16 : nullptr);
                else
                    ok = substitute_tile<1>(T, Lkk, Np, Wk, sf + 4 * kd, want, jb.out, smem, X,
                                            st.stamps ? st.stamps + (int64_t)blockIdx.x * 16 : nullptr);
                if ((tid & 63) == 0 && !ok) atomicCAS(info_b, 0, (int)0x80000000);
                SMALL_STAMP(4);
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
    SMALL_STAMP(5);
    if (tid == 0) {                                                // the last workgroup out closes the step
        const int f = __hip_atomic_fetch_add(st.hdr + 4, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (f == (int)gridDim.x - 1) {
            __hip_atomic_store(st.hdr + 4, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st.hdr + 3, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

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
        geti("VOLT_BATCH_LAD", t.batch_lad);
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
    t.small_nmax = 0;                                        // the one-launch steps: tuned for, and (batched step) placed on, the full chip
    t.long_on = 0;
    t.batch = 0;
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
    hipLaunchKernelGGL(small_step_kernel, dim3(B * small_pieces(n)), dim3(256), pad, (hipStream_t)stream, A, Winv, Y, info, Np,
                       B, src, red, st, tl);
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
    const LongState st{base, base + SMALL_HDR, tab, tab + pd->nitems, eslab, g_small_stamps, pd->xcd_from, tunables().long_split};
    // one workgroup per CU (16 KB of LDS padding): a pivot chain that shares its CU runs 1.5 - 3x slower
    const unsigned pad = tunables().long_pad ? 16 * 1024 : 0;
    hipLaunchKernelGGL(long_step_kernel, dim3(pd->nitems), dim3(256), pad, s, A, Winv, Y, info, Np, src, red, st, tl);
    VOLT_LAUNCH_CHECK();
    return 1;
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
