// The whole batched MLL + gradient step (or factorisation) in ONE launch (round 5).
//   reference: ONE loop body per training step, voltron/train_utils.py:243-254 (loss = -mll(model(x), y); backward()).
//
// The launch-per-column schedules of chol.hip give a step of B matrices of n block columns n + 1 launches, each as long as
// its longest tile and ended by a tail in which most CUs idle; two stream groups hide part of that, a hipGraph capture
// cannot (one group), and 8 .. 16 matrices at N = 4096 or 64 at N = 2048 are bounded by exactly these boundaries.  Here the
// pieces of the host's topologically ordered list (batch_sched.h) -- diagonal tiles, look-ahead tiles, two-phase panel tiles,
// rows of the triangular inverse -- are pulled BY TICKET by a grid of resident workgroups (round 6, common.h "who runs which
// piece": nothing is assumed about where or in what order workgroups start); the tile bodies are the ones every other
// schedule runs (tiles.h), so the arithmetic, and with it every bit of the result, is the launch-per-column path's.  What is new:
//   * hand-offs by PROGRESS WORDS per matrix in the caller's scratch (cleared by batch_begin_kernel ahead of the launch):
//       rowp[i]  = block columns of block row i of L that are complete (P(i,k) stores k + 1 behind its tile)
//       tcol[j]  = tiles of block column j of X = L^-1 that are complete (TD(j) stores 1, T(i,j) stores i - j + 1)
//       la[k]    = the look-ahead part of A[k,k] is parked
//       W_k's first word as before (diag_body)
//   * tiles CHASE their inputs (common.h): a panel tile (i,k) follows rowp[k] and rowp[i], a tile (i,j) of the inverse
//     rowp[i] and tcol[j], asking once per 128-wide K segment and only while its inputs are incomplete.  A tile dispatched
//     a block column early has all but its last K block behind it when that column completes.
//   * panel and inverse tiles are stored WRITTEN THROUGH (sc1, non-temporal) and published behind the storing waves' own
//     drain -- no L2-wide write-back per tile (what made per-slice releases slower the more slices there were: chol.hip,
//     slab_dump).  Look-ahead tiles, diagonal tiles of the inverse and W_k (one per matrix and block column) keep the
//     release fence.
// The table (16 bytes per piece) and the progress words live in the caller's workspace (volt_mll_workspace_init_f32 /
// volt_potrf_workspace_init_f32 copy the table there once); every workgroup checks the table's header against the
// shape it was launched for and reports a region that does not hold it (info = INT_MIN + 1) instead of following it.
#include "common.h"
#include "tiles.h"
#include "host.h"
#include "batch_sched.h"
#include "../../include/volt_hip.h"
#include "../../include/volt_hip_tune.h"
#include <algorithm>
#include <array>
#include <map>
#include <mutex>
#include <string.h>
#include <vector>

namespace volt {

// clears what a step's hand-offs are made of: the first word of every W block, info, the progress words
__global__ void batch_begin_kernel(float* __restrict__ Winv, int nflags, int* __restrict__ info, int ninfo,
                                   int* __restrict__ prog, int nprog) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nflags) reinterpret_cast<int*>(Winv)[(int64_t)i * TS * TS] = 0;
    if (i < ninfo) info[i] = 0;
    for (int c = i; c < nprog; c += gridDim.x * blockDim.x) prog[c] = 0;
}

// (word_ge / batch_wait / batch_publish_*: common.h -- shared with the fp64 one-launch step, batch64_step.hip)
constexpr int AUX_WT = AUX_SC1;                // sc1: written through at agent scope (never nt on a hand-off: MI355X_MICROARCH.md)

// First diagonal tile straight from the caller's K (+ sigma2 / jitter on the diagonal, identity in the padding) into the
// pivot image: no prepared copy of block column 0, no launch ahead of the step for it.
__device__ __forceinline__ void diag0_image(const KSource& src, int b, float* smem) {
    const float add = (src.sigma2 ? src.sigma2[b] : 0.f) + src.jitter;
    const float* Kb = src.K + (int64_t)b * src.bsk;
    // sixteen loads in flight per thread (clamped addresses; padding, diagonal and upper triangle selected afterwards) -- one
    // element at a time this was 64 round trips in front of the first pivot of every matrix of the step
    const int N = src.N;
#pragma unroll 1
    for (int e0 = threadIdx.x; e0 < TS * TS; e0 += 16 * NT) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = e0 + u * NT, r = e >> 7, c = e & 127;
            v[u] = Kb[(int64_t)(r < N ? r : N - 1) * src.ldk + (c < N ? c : N - 1)];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = e0 + u * NT, r = e >> 7, c = e & 127;
            float x = (r < N && c < N) ? v[u] : 0.f;
            if (r == c) x = (r < N) ? x + add : 1.f;
            smem[r * DT + c] = (c <= r) ? x : 0.f;
        }
    }
}

// AL(c): block c of z = Y'r -- the sum of the z-partials of row c of the inverse, in sum_zpart_kernel's order -- and
// alpha's partial sums from block column c of Y, apart[b][c][row] = Y[row, block c] . z_c for the rows of block rows
// 0 .. c (rowpair_dot: the sums y_times_z_kernel forms).  A stream over (c + 1) tiles the workgroups of this launch have
// just written.
__device__ __forceinline__ void alpha_item(const float* __restrict__ Y, const TriReduce& red, float* __restrict__ z,
                                           float* __restrict__ apart, int Np, int b, int c, int chunk) {
    const int n = Np / TS, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
    f32x4 zc = {0.f, 0.f, 0.f, 0.f};
    {   // eight loads in flight, added in the order of ever (jb ascending): one at a time this was up to 32 round trips per piece
        const float* zp = red.zpart + (int64_t)b * n * Np + c * TS + 4 * l31;
        int jb = 0;
        for (; jb + 8 <= c + 1; jb += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(zp + (int64_t)(jb + u) * Np);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) zc[e] += v[u][e];
        }
        for (; jb <= c; ++jb) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(zp + (int64_t)jb * Np);
#pragma unroll
            for (int e = 0; e < 4; ++e) zc[e] += v[e];
        }
    }
    if (chunk == 0 && wave == 0 && h == 0) *reinterpret_cast<f32x4*>(z + (int64_t)b * Np + c * TS + 4 * l31) = zc;
    const float* Yc = Y + (int64_t)b * Np * Np + c * TS + 4 * l31 + (int64_t)h * Np;
    float* dst = apart + ((int64_t)b * n + c) * Np;
    const int r0 = chunk * BATCH_ALPHA_ROWS;
    const int r1 = (r0 + BATCH_ALPHA_ROWS < (c + 1) * TS) ? r0 + BATCH_ALPHA_ROWS : (c + 1) * TS;
    // a wave takes 32 consecutive rows per pass as 16 row pairs (r + 2 u + h), all sixteen loads in flight: the stream is
    // latency-bound otherwise (four in flight: 280 us per piece; thirty-two: no faster than sixteen)
    for (int r = r0 + 32 * wave; r < r1; r += 128) {
        f32x4 y[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) y[u] = *reinterpret_cast<const f32x4*>(Yc + (int64_t)(r + 2 * u) * Np);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float lo, hi;
            rowpair_dot(y[u], zc, lo, hi);
            if (lane == 0) {
                dst[r + 2 * u] = lo;
                dst[r + 2 * u + 1] = hi;
            }
        }
    }
}

// tuning only (volt_tune_batch_stamps): 8 int64 per workgroup -- s_memrealtime at entry and exit, the hardware id, and for
// the two-phase tiles the time the pipeline is entered and left
static long long* g_batch_stamps = nullptr;

// (who runs which piece -- the pullers, their queues and BATCH_QWORDS: common.h)
struct BatchArgs {
    float* A; float* Winv; float* Y; int* info;
    int Np, B;
    KSource src;
    TriReduce red;
    const int4* tab;
    int* prog;
    int pstride;
    float* zvec; float* apart;
    long long* stamps;
};

// LDS of the step kernel
static __shared__ __attribute__((aligned(16))) float g_smem[2 * STAGE_FLOATS];
static __shared__ float g_srv[TS];                           // a tile of the inverse: the residuals of its block row (trtri_reduce)
static __shared__ int g_piece;                               // the piece thread 0 pulled
static __shared__ int g_ahead[2];                            // [iteration parity] 1: g_piece / g_desc were fetched AHEAD, during the last tile's epilogue
static __shared__ int4 g_desc;                               // ... that piece's table entry

// One piece of the list.  LOCAL: the hand-offs of this piece's matrix all happen under this workgroup's L2 (see above):
// plain (non-temporal) stores, acknowledged by the L2, instead of write-through to memory -- the word of a tile follows its
// stores after ~1 us instead of ~10.
template <bool FROMK, bool LOCAL>
__device__ __forceinline__ void batch_piece(const BatchArgs a, const int w, const bool ahead, const int par, BatchPull& pull,
                                            int* qw, int per_queue) {
    float* const smem = g_smem;
    float* const srv = g_srv;
    float* const A = a.A; float* const Winv = a.Winv; float* const Y = a.Y; int* const info = a.info;
    const int Np = a.Np;
    const KSource& src = a.src;
    const TriReduce& red = a.red;
    long long* const stamps = a.stamps;
    const int n = Np / TS;
    int4 d = ahead ? g_desc : a.tab[BATCH_HDR + w];          // (fetched during the last tile's epilogue, or now)
    // (the descriptor arrives in vector registers: said to be uniform HERE, or the compiler may turn every scalar computation
    // that follows from it -- tile addresses, buffer descriptors -- into per-lane code with a waterfall loop around each load)
    d.x = __builtin_amdgcn_readfirstlane(d.x);
    d.y = __builtin_amdgcn_readfirstlane(d.y);
    d.z = __builtin_amdgcn_readfirstlane(d.z);
    if (stamps && threadIdx.x == 0) stamps[(int64_t)w * 8 + 5] = __builtin_amdgcn_s_memrealtime();
    const int kind = d.x & 7, b = d.x >> 3;
    int* rowp = a.prog + (int64_t)b * a.pstride;
    int* tcol = rowp + n;
    int* la = tcol + n;
    int* info_b = info + b;
    const bool usek = FROMK && src.K != nullptr;

    if (kind == BK_DIAG) {
        const int k = d.y;
        // The pivot chain is what a block column waits for, and here it shares its CU with a tile that issues MFMAs and LDS
        // reads from every SIMD: left at the default priority its dependent readlane / FMA chain took 60 - 150 us instead of
        // the 30 it takes alone (8 x 4096, profiles/r05 stamps).  Raised above the co-resident tile's waves it is issued the
        // cycle it is ready -- and being a dependent chain it leaves almost every issue slot to the other tile anyway.
        __builtin_amdgcn_s_setprio(3);
        if (k == 0) {
            if (usek) {
                diag0_image(src, b, smem);
                diag_body<false, false, LOCAL>(A, Winv, info, Np, 0, b, smem, nullptr, true);
            } else {
                diag_body<false, false, LOCAL>(A, Winv, info, Np, 0, b, smem, nullptr, false);
            }
            return;
        }
        // L[k,k-1] (and with it all of row k) is there; k >= 2: the look-ahead part of A[k,k] is parked
        batch_wait<LOCAL>(rowp + k, k, k >= 2 ? la + k : nullptr, 1, info_b);
        if (stamps && threadIdx.x == 0) stamps[(int64_t)w * 8 + 3] = __builtin_amdgcn_s_memrealtime();
        if (k >= 2) update_body<FROMK>(A, Np, k, k, k - 1, k, false, b, src, smem, true);
        else update_body<FROMK>(A, Np, 1, 1, 0, 1, true, b, src, smem, true);
        if (stamps && threadIdx.x == 0) stamps[(int64_t)w * 8 + 4] = __builtin_amdgcn_s_memrealtime();
        diag_body<false, false, LOCAL>(A, Winv, info, Np, k, b, smem, nullptr, true);
        return;
    }
    if (kind == BK_LOOKAHEAD) {
        const int k = d.y;                                   // A[k+1,k+1] -= sum_{m<k} L[k+1,m] L[k+1,m]^T, chasing row k+1
        Chase ch;
        ch.p0 = ch.p1 = rowp + k + 1;
        bool ok = true;
        update_body<FROMK, 0, true, LOCAL>(A, Np, k + 1, k + 1, 0, k, true, b, src, smem, false, &ch, &ok);
        if (!ok && (threadIdx.x & 63) == 0) atomicCAS(info_b, 0, (int)0x80000000);
        batch_publish_release<LOCAL>(la + k + 1, 1);
        return;
    }
    if (kind == BK_TRTRI_DIAG) {
        const int i = d.y;
        batch_wait<LOCAL>(reinterpret_cast<const int*>(Winv + ((int64_t)b * n + i) * TS * TS), 1, nullptr, 0, info_b);
        trtri_diag_body(Winv, Y, Np, i, b, red, smem);
        batch_publish_release<LOCAL>(tcol + i, 1);
        return;
    }
    if (kind == BK_ALPHA) {
        const int c = d.y;
        if (threadIdx.x < 64) {                              // row c of the inverse is complete: tcol[j] >= c - j + 1, j <= c
            bool ok = true;
            for (int j = 0; j <= c; ++j) ok = wait_word_ge<LOCAL>(tcol + j, c - j + 1) && ok;   // (wave-uniform addresses)
            if (!ok && threadIdx.x == 0) atomicCAS(info_b, 0, (int)0x80000000);
            acquire_unless_local<LOCAL>();
        }
        __syncthreads();
        alpha_item(Y, red, a.zvec, a.apart, Np, b, c, d.z);
        return;
    }
    // ---- the two-phase tiles
    TriJob jb;
    int* word;
    int val;
    if (kind == BK_PANEL) {
        const int i = d.y, k = d.z;
        jb = panel_job<FROMK>(A, Winv, Np, i, k, b, src);
        jb.t.ch.p0 = rowp + k;                               // X = L[k, 0 ..]
        jb.t.ch.p1 = rowp + i;                               // Z = L[i, 0 ..]
        word = rowp + i;
        val = k + 1;
    } else {
        const int i = d.y, j = d.z;
        jb = trtri_job(A, Winv, Y, Np, i, j, b);
        if (red.rpad && threadIdx.x < TS) srv[threadIdx.x] = red.rpad[(int64_t)b * Np + j * TS + threadIdx.x];   // (read behind the pipeline's barriers)
        jb.t.flag = reinterpret_cast<const int*>(jb.t.W);    // W_i comes from a workgroup of this launch too
        jb.t.ch.p0 = rowp + i;                               // X = L[i, j ..]: block m is L[i, j + m]
        jb.t.ch.base0 = j;
        jb.t.ch.p1 = tcol + j;                               // Z = Y[j, j ..]: block m is tile (j + m, j) of the inverse
        word = tcol + j;
        val = i - j + 1;
    }
    f32x16 T[4], O[4];
    ChasePre pre = {0, 0};
    if (jb.t.n1 > 0) pre = chase_issue<LOCAL>(jb.t.ch);             // the polls go out ahead of the input tile's loads: one round trip
    job_t0(jb, T);
    if (stamps && threadIdx.x == 0) stamps[(int64_t)w * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    const bool ok = tri_tile_run<true, LOCAL>(jb.t, T, O, smem, &pre);
#ifndef VOLT_NO_EPILOGUE_PRIO
    // The epilogue -- reductions, the tile through LDS, the drain -- is a few hundred instructions that, beside a co-resident
    // tile in its K loop, took 9 + 6 us instead of 1 + 1.3 (profiles/r05/batch_stamps.txt: 64 x 4096 against 8 x 4096): raised
    // above that tile's waves they are issued when they are ready, and the slot turns over sooner.
    __builtin_amdgcn_s_setprio(3);
#endif
    if (stamps && threadIdx.x == 0) stamps[(int64_t)w * 8 + 4] = __builtin_amdgcn_s_memrealtime();
    if (!ok && (threadIdx.x & 63) == 0) atomicCAS(info_b, 0, (int)0x80000000);   // a hand-off timed out: internal error
    // The NEXT piece's ticket goes out here, ahead of the epilogue: the atomic's round trip (~1 us beside busy neighbours) rides
    // under the reductions and the tile's stores, whose drain (tri_store_lds ends on vmcnt(0)) also brings it back; its table
    // entry is then requested ahead of the publish's drain + barrier.  At the top of the loop both are there -- a tile every
    // ~40 us (64 x 2048) to ~160 us (64 x 4096) per puller otherwise starts with two dependent round trips.  Taking a ticket
    // early is safe: its holder runs it right after this piece, and a ticket only ever waits for smaller ones.
    int t_next = -1;
    if (threadIdx.x == 0 && pull.q >= 0) t_next = __hip_atomic_fetch_add(qw + 32 * pull.q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the reductions of a tile of the inverse first (they read the accumulators and use LDS behind the tile image), then
    // the tile: its stores are the last thing before the drain
    if (jb.i >= 0 && red.rpad) trtri_reduce<!LOCAL>(O, Np, jb.i, jb.j, jb.b, red, smem + TS * WLD, srv);
    if (stamps && threadIdx.x == 0) stamps[(int64_t)w * 8 + 6] = __builtin_amdgcn_s_memrealtime();
    tri_store_lds<LOCAL ? 0 : AUX_WT>(O, jb.out, Np, smem);
    if (stamps && threadIdx.x == 0) stamps[(int64_t)w * 8 + 7] = __builtin_amdgcn_s_memrealtime();
    int4 d_next = {0, 0, 0, 0};
    int w_next = -1;
    if (threadIdx.x == 0 && pull.q >= 0) {
        if (t_next < per_queue) {
            w_next = LOCAL ? 8 * t_next + pull.q : t_next;
            d_next = a.tab[BATCH_HDR + w_next];
        } else {
            pull.q = -1;                                     // this queue is dry: the loop's own pull looks for another
        }
    }
    batch_publish_wt<LOCAL>(word, val);
    if (threadIdx.x == 0 && w_next >= 0) {                   // (the loop's barriers order these against the other waves' reads)
        g_ahead[par ^ 1] = 1;
        g_piece = w_next;
        g_desc = d_next;
    }
}

template <bool FROMK, bool LOCAL>
__global__ __launch_bounds__(256, 2) void batch_step_kernel(BatchArgs a, int check, int npieces, int xskew, int xdrop) {
    {   // the caller's scratch must hold the table this launch was sized for: anything else is reported, never followed
        const int4 h0 = a.tab[0];
        if (h0.x != BATCH_MAGIC || h0.y != a.B || h0.z != a.Np / TS || h0.w != check) {
            if (blockIdx.x == 0)
                for (int b = threadIdx.x; b < a.B; b += NT) a.info[b] = (int)0x80000001;
            return;
        }
    }
    int* const qw = a.prog + (int64_t)a.B * a.pstride;       // the queue words sit behind the progress words
    const int hw = hw_xcc_id();
    if (LOCAL && ((xdrop >> hw) & 1)) return;
    const int xcc = (hw + xskew) & 7;
    const int per_queue = LOCAL ? npieces / 8 : npieces;
    BatchPull pull;
    // The pullers' loop -- written so that the optimiser does NOT see a loop.  As `for (;;)` everything a piece kind derives
    // from the thread index and the arguments (lane offsets, strides, descriptors: of all six kinds) is hoisted out as loop
    // invariant, kept alive across the tile pipelines and spilled (400 - 1000 VGPRs); as a called function the body saves 112
    // callee-saved registers per piece.  A second way into the cycle (never taken: the host passes xdrop without bit 30) makes
    // it irreducible: no loop pass touches it, and the body compiles as round 5's one-piece kernel did (254 VGPRs, no scratch).
    int w, ahead = 0, par = -1;                              // par: parity of the iteration (-1: the first, nothing fetched ahead)
    if (xdrop & 0x40000000) {
        w = xskew | (int)0x80000000;
        goto piece;
    }
pull_next:
    __syncthreads();                                         // the last piece's LDS traffic (and its read of g_piece) is over
    if (threadIdx.x == 0 && !(par >= 0 && g_ahead[par])) g_piece = batch_next_piece<LOCAL>(pull, qw, xcc, per_queue);
    __syncthreads();
    w = g_piece;
    ahead = par >= 0 ? g_ahead[par] : 0;
    par = par < 0 ? 0 : par;
    if (threadIdx.x == 0) g_ahead[par ^ 1] = 0;              // (nobody reads the other parity before the next two barriers; a tile's epilogue may set it)
piece:
    w = __builtin_amdgcn_readfirstlane(w);
    if (w < 0) return;
    __builtin_amdgcn_s_setprio(0);
    if (a.stamps && threadIdx.x == 0) {
        a.stamps[(int64_t)w * 8] = __builtin_amdgcn_s_memrealtime();
        a.stamps[(int64_t)w * 8 + 2] = ((long long)hw << 32) | (unsigned)__builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);
    }
    par = __builtin_amdgcn_readfirstlane(par < 0 ? 0 : par);
    batch_piece<FROMK, LOCAL>(a, w, __builtin_amdgcn_readfirstlane(ahead) != 0, par, pull, qw, per_queue);
    par ^= 1;
    if (a.stamps && threadIdx.x == 0) a.stamps[(int64_t)w * 8 + 1] = __builtin_amdgcn_s_memrealtime();
    goto pull_next;
}

}  // namespace volt

using namespace volt;

// Where the one launch replaces the launch-per-column schedules: the measured crossovers of profiles/r05/batch_gate_sweep.txt (re-swept
// around the borders with round 6's pullers: profiles/r06/batch_gate_sweep.txt)
// (ms launch-per-column / ms one-launch over B = 2 .. 96, N = 1024 .. 4096, for the gradient step and for the factorisation
// alone).  (B, n, inverse?) only: the gate is part of the shape's identity -- volt_*_workspace_bytes / _init and the step
// agree on it.  Short series of few matrices stay with the short-series one-launch step (small_step_kernel), one long series
// with long_step_kernel: they are tried first (mll.hip).
bool volt_internal_batch_applies(int B, int n, int has_y) {
    const Tunables& tn = tunables();
    if (tn.batch <= 0 || B < 1 || n < 2 || B > 65535) return false;
    if (tn.batch >= 2) return true;
    if (has_y) {
        if (n >= 20) return B >= 2;                          // N = 3072, 4096: 1.01 - 1.44 x at every batch size from 2 on
        if (n >= 16) return B >= 4;                          // N = 2048: 4 matrices 1.05 x with the pullers (round 6 sweep; 2 - 3: 0.94 - 0.99)
        if (n >= 10) return B >= 6;                          // N = 1536: 1.04 - 1.42 x (2 - 4 matrices: 0.97)
        if (n >= 8) return B >= 20;                          // N = 1024: 1.07 - 1.48 x
        return false;
    }
    if (n >= 28) return B >= 2;                              // the factorisation alone: 1.02 - 1.74 x
    if (n >= 20) return B >= 3;
    if (n >= 10) return B >= 8;                              // (10, 12 matrices of N = 1536: a tie)
    if (n >= 8) return B >= 24;
    return false;
}

bool volt_internal_batch_first() { return tunables().batch >= 3; }   // tuning: ahead of the short- / long-series steps

// look-ahead tiles listed this many block columns early (batch_sched.h): only while the forward waiters are a small part of
// the 512 resident workgroups
static int batch_lad(int B) {
    int lad = tunables().batch_lad;
    while (lad > 0 && 2 * lad * B > 128) --lad;
    return lad;
}

static size_t batch_table_bytes(int B, int n, bool has_y) {
    return ((size_t)(BATCH_HDR + batch_count(B, n, has_y)) * sizeof(BatchItem) + 255) & ~(size_t)255;
}
static size_t batch_prog_bytes(int B, int n) {                 // progress words of every matrix, then the queue words
    return (((size_t)B * batch_pstride(n) + BATCH_QWORDS) * sizeof(int) + 255) & ~(size_t)255;
}

size_t volt_internal_batch_bytes(int B, int n, int has_y) {
    if (!volt_internal_batch_applies(B, n, has_y)) return 0;
    return batch_table_bytes(B, n, has_y != 0) + batch_prog_bytes(B, n);
}

// Order of a block column's tiles in the list (batch_sched.h).  Round 6: with the pullers a queue's order is
// free, and WINDOWS of 32 positions x 2 groups of 8 matrices (order 54: an XCD's 64 pullers then work on 2 of its matrices at a
// time, not 8) cut the step's HBM reads 79.4 -> 64.0 GB at 64 x 4096 (84.1 -> 68.7 GB with the writes; every window shape tried
// plateaus at 68.5: profiles/r06/order_traffic.txt) at the SAME time (22.06 against 22.10 ms) -- but cost 0.7 % at 32 x 4096 and 2 %
// at 64 x 2048, where a window is most of a block column and the chain waits for it.  So: windows from 48 matrices of 28 block
// columns on, positions matrix-innermost elsewhere.  VOLT_BATCH_ORDER >= 0 (tuning) forces one order everywhere.
static int batch_order_for(int B, int n) {
    const int forced = tunables().batch_order;
    if (forced >= 0) return forced;
    return (B >= 48 && n >= 28 && (B & 7) == 0) ? 54 : 0;
}

// the word every piece of a table carries: what the table was built for
static int batch_check_word(int B, int n, bool has_y) {
    const uint32_t w = (uint32_t)BATCH_MAGIC ^ ((uint32_t)B * 0x01000193u) ^ ((uint32_t)n << 20) ^ (has_y ? 0x40000000u : 0u) ^
                       ((uint32_t)batch_order_for(B, n) * 2654435761u) ^ ((uint32_t)batch_lad(B) << 16);
    return (int)w;
}

// The table, built once per (B, n, inverse?, order) in pinned host memory and kept for the life of the library (host
// memory only, like the balanced schedules' tables); volt_*_workspace_init copies it into the caller's scratch.
struct BatchTable {
    int4* items = nullptr;
    size_t bytes = 0;
};
static const BatchTable* get_batch_table(int B, int n, bool has_y) {
    static std::mutex mu;
    static std::map<std::array<int, 4>, BatchTable*> cache;
    const int order = batch_order_for(B, n);
    const std::array<int, 4> key{B, n, has_y ? 1 : 0, order * 16 + batch_lad(B)};
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    std::vector<BatchItem> items;
    batch_build(B, n, has_y, order, items, batch_lad(B));
    BatchTable* bt = new BatchTable;
    static_assert(sizeof(BatchItem) == sizeof(int4), "items are read as int4");
    bt->bytes = (BATCH_HDR + items.size()) * sizeof(BatchItem);
    if ((int64_t)items.size() != batch_count(B, n, has_y) ||
        hipHostMalloc((void**)&bt->items, bt->bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        delete bt;
        bt = nullptr;
    } else {
        memset(bt->items, 0, BATCH_HDR * sizeof(BatchItem));
        bt->items[0] = int4{BATCH_MAGIC, B, n, batch_check_word(B, n, has_y)};
        for (BatchItem& it : items) it.pad = batch_check_word(B, n, has_y);
        memcpy(bt->items + BATCH_HDR, items.data(), items.size() * sizeof(BatchItem));
    }
    cache[key] = bt;
    return bt;
}

int volt_internal_batch_install(void* state, size_t bytes, int B, int n, int has_y, void* stream) {
    if (!state || bytes < volt_internal_batch_bytes(B, n, has_y) || !volt_internal_batch_applies(B, n, has_y)) return 0;
    const BatchTable* bt = get_batch_table(B, n, has_y != 0);
    if (!bt) return (int)hipErrorOutOfMemory;
    hipError_t e = hipMemcpyAsync(state, bt->items, bt->bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    return e != hipSuccess ? (int)e : 0;
}

// One step: clear the hand-off words, then the one launch.  K != nullptr: tiles read their input from K (A receives L);
// K == nullptr: A holds the input (volt_potrf_f32).  Y == nullptr: the factorisation alone.  e0 / e1 (optional): events
// recorded around the step kernel on `stream` (bench.py's roofline leg).  Returns 1 when the step was enqueued, 0 when
// the shape is not this schedule's (nothing enqueued), a HIP error otherwise.
int volt_internal_batch_step(const float* K, int64_t ldk, int64_t bsk, const float* sigma2, float jitter, float* A,
                             float* Winv, float* Y, int* info, const float* rpad, float* zpart, float* frob, int B, int N,
                             float* z, float* apart, void* state, size_t state_bytes, void* stream, hipEvent_t e0,
                             hipEvent_t e1) {
    const int Np = volt_padded_n(N), n = Np / TS;
    const bool has_y = Y != nullptr;
    if (!state || !volt_internal_batch_applies(B, n, has_y) || state_bytes < volt_internal_batch_bytes(B, n, has_y)) return 0;
    if (has_y && (!z || !apart)) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int4* tab = reinterpret_cast<const int4*>(state);
    int* prog = reinterpret_cast<int*>(reinterpret_cast<char*>(state) + batch_table_bytes(B, n, has_y));
    const int pstride = batch_pstride(n), nprog = B * pstride + BATCH_QWORDS, nflags = B * n;
    int blocks = (std::max(std::max(nflags, B), nprog) + 255) / 256;
    if (blocks > 256) blocks = 256;
    if (blocks * 256 < std::max(nflags, B)) blocks = (std::max(nflags, B) + 255) / 256;
    hipLaunchKernelGGL(batch_begin_kernel, dim3(blocks), dim3(256), 0, s, Winv, nflags, info, B, prog, nprog);
    const int check = batch_check_word(B, n, has_y);
    // eight queues, one per XCD, when the matrices divide among them evenly (the pullers read their XCC id: batch_step_kernel)
    const bool local = (B & 7) == 0 && tunables().batch_local != 0 && tunables().xccs == 8;
    const KSource src{K, ldk, bsk, sigma2, jitter, N};
    const TriReduce red{has_y ? rpad : nullptr, zpart, frob, N};
    const int64_t npieces = batch_count(B, n, has_y);
    // Few matrices: ONE workgroup per CU (16 KB of dynamic LDS padding).  With two, the diagonal tile -- the latency chain a
    // block column waits for -- shares its CU's LDS and issue slots with a tile in its K loop and takes 80 us instead of 32
    // (8 x 4096 stamps, profiles/r05); alone on the CU it runs at its own pace, and the tiles lose only the few percent
    // that a second resident tile adds to the MFMA duty.
    const int spread = tunables().batch_spread + (n < 32 ? 9 * (32 - n) * tunables().cus / 256 : 0);   // (measured crossovers: host.h)
    const unsigned pad = (int64_t)B * (n + 1) <= spread ? 16 * 1024 : 0;
    // as many pullers as the chip holds at once (nothing depends on the number: a puller that starts late finds the queues
    // further on, or dry)
    const int64_t slots = (int64_t)tunables().cus * (pad ? 1 : 2) * std::max(1, tunables().batch_pullers);
    const unsigned grid = (unsigned)std::min<int64_t>(npieces, tunables().batch_pullers > 0 ? slots : npieces);
    const BatchArgs args{A, Winv, Y, info, Np, B, src, red, tab, prog, pstride, z, apart, g_batch_stamps};
    const int xskew = tunables().batch_xskew, xdrop = tunables().batch_xdrop;
    if (e0 && hipEventRecord(e0, s) != hipSuccess) return (int)hipGetLastError();
#define VOLT_BATCH_LAUNCH(FK, LC) \
    hipLaunchKernelGGL((batch_step_kernel<FK, LC>), dim3(grid), dim3(256), pad, s, args, check, (int)npieces, xskew, xdrop)
    if (K && local) VOLT_BATCH_LAUNCH(true, true);
    else if (K) VOLT_BATCH_LAUNCH(true, false);
    else if (local) VOLT_BATCH_LAUNCH(false, true);
    else VOLT_BATCH_LAUNCH(false, false);
#undef VOLT_BATCH_LAUNCH
    if (e1 && hipEventRecord(e1, s) != hipSuccess) return (int)hipGetLastError();
    hipError_t e = hipGetLastError();
    return e != hipSuccess ? (int)e : 1;
}

extern "C" {

// What the library found the device to be, and the gates that follow from it (host.h): out[0] CUs, [1] XCDs, [2] slots the
// balanced schedule plans for, [3] / [4] plain / split launches up to this many workgroups run one per CU, [5] launches
// below this many workgroups run as one stream group, [6] one-launch steps enabled (short series | one long series << 1 |
// batched << 2).  No GPU needed (a process without a device reports the full-chip defaults).
int volt_topology_describe(int* out) {
    if (!out) return -1;
    const Tunables& tn = tunables();
    out[0] = tn.cus;
    out[1] = tn.xccs;
    out[2] = tn.sched_g;
    out[3] = tn.plain_spread;
    out[4] = tn.split_spread;
    out[5] = tn.group_gate;
    out[6] = (tn.small_nmax > 0 ? 1 : 0) | (tn.long_on ? 2 : 0) | (tn.batch > 0 ? 4 : 0);
    return 0;
}

int volt_tune_batch_stamps(long long* stamps) {
    g_batch_stamps = stamps;
    return 0;
}

// Host only (no GPU): the piece list of the one-launch batched step, in grid order -- items [max_items][4] int32
// {kind | b << 3, row, col, 0} (batch_sched.h).  Returns the number of pieces, -1 bad argument, -2 max_items too small.
int volt_batch_describe(int B, int n, int has_y, int order, int* items, int max_items) {
    // order 0 .. 15: (order & 1) = matrix innermost / matrix-major, order >> 1 = look-ahead tiles listed that many columns early;
    // 1000 + w: the windowed order w of batch_sched.h (what large batches run: batch_order_for);  -1: the order the step picks
    if (order == -1) order = 1000 + batch_order_for(B, n);
    if (B < 1 || n < 1 || order < 0 || (order > 15 && order < 1000) || order > 1255) return -1;
    std::vector<BatchItem> it;
    if (order >= 1000) batch_build(B, n, has_y != 0, order - 1000, it, 0);
    else batch_build(B, n, has_y != 0, order & 1, it, order >> 1);
    if ((int64_t)it.size() != batch_count(B, n, has_y != 0)) return -1;
    if (items) {
        if ((int)it.size() > max_items) return -2;
        memcpy(items, it.data(), it.size() * sizeof(BatchItem));
    }
    return (int)it.size();
}

}  // extern "C"
