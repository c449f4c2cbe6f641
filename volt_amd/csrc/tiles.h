// Tile bodies of the blocked factorisation / triangular inverse (device code only, gfx950): the 2x2-wave update of a
// diagonal tile, the 128-pivot diagonal block (readlane idiom) with its inverse, the two-phase panel / trtri tile set-up,
// the trtri epilogue reductions.  Shared by every step kernel (chol.hip: one launch per block column, split-K, balanced,
// short and long one-launch steps; batch_step.hip: the whole batched step in one launch).
#pragma once
#include "common.h"

namespace volt {

// ----------------------------------------------------------------------------- panel update (diagonal tiles)
// C tiles of block columns >= 1 come straight from the caller's K (+ sigma2/jitter on the diagonal, identity in the
// padding) instead of a prepared copy in A, which removes the K -> A copy pass for every block column but the first.
struct KSource {
    const float* K;          // nullptr: the working matrix A already holds the input (volt_potrf_f32)
    int64_t ldk, bsk;
    const float* sigma2;
    float jitter;
    int N;
};

// Element (gi, gj) of the input matrix: from K (+ sigma2/jitter on the diagonal, identity in the padding), or from A.
__device__ __forceinline__ float input_elem(const KSource& src, const float* Kb, float add, const float* Ab, int Np,
                                            bool usek, int gi, int gj) {
    if (usek) {
        float v = (gi < src.N && gj < src.N) ? Kb[(int64_t)gi * src.ldk + gj] : 0.f;
        if (gi == gj) v = (gi < src.N) ? v + add : 1.f;
        return v;
    }
    return Ab[(int64_t)gi * Np + gj];
}

// Diagonal tile (kb,kb) of the working matrix:  C <- C0 - sum_{m = kb0}^{kb1-1} L[kb, m] L[kb, m]^T, where C0 is the
// tile as it stands in A, or -- when `fromk` -- the caller's K.  kb0 > 0 continues an update begun by an earlier
// launch (diagonal look-ahead, see factor_step_kernel).  The C tile is loaded NEGATED straight into the accumulators
// before the K loop (acc = -C + sum, C <- -acc): nothing is held back for an epilogue read-modify-write, and no
// prefetch registers are carried through the loop.
template <bool FROMK, int ABL = 0, bool CHASE = false, bool LOCALP = false>
__device__ __forceinline__ void update_body(float* __restrict__ A, int Np, int rowblk, int colblk, int kb0, int kb1,
                                            bool fromk, int b, const KSource& src, float* smem,
                                            bool to_image = false, const Chase* ch = nullptr, bool* ch_ok = nullptr) {
    float* Ab = A + (int64_t)b * Np * Np;
    const float* Arows = Ab + (int64_t)rowblk * TS * Np + (int64_t)kb0 * TS;   // L[rowblk, kb0:kb1]
    const float* Brows = Ab + (int64_t)colblk * TS * Np + (int64_t)kb0 * TS;   // L[colblk, kb0:kb1]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    float* C = Ab + (int64_t)rowblk * TS * Np + (int64_t)colblk * TS;
    f32x16 acc[4];
    const bool usek = FROMK && fromk;                      // workgroup-uniform
    const float add = usek ? ((src.sigma2 ? src.sigma2[b] : 0.f) + src.jitter) : 0.f;
    const float* Kb = usek ? src.K + (int64_t)b * src.bsk : nullptr;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r = wr * 64 + tm * 32 + accrow(q, lane);
                const int c = wc * 64 + tn * 32 + (lane & 31);
                if (ABL & 1) acc[tm * 2 + tn][q] = 0.f;         // ablation (tuning only): no C load
                else acc[tm * 2 + tn][q] = -input_elem(src, Kb, add, Ab, Np, usek, rowblk * TS + r, colblk * TS + c);
            }
    gemm_nt_128<0, CHASE, LOCALP>(Arows, Np, Brows, Np, (kb1 - kb0) * (TS / BK), acc, smem, ch, ch_ok);
    if (ABL & 2) {                                                 // ablation (tuning only): one store per thread
        float sum = 0.f;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int q = 0; q < 16; ++q) sum += acc[t4][q];
        C[(int64_t)(threadIdx.x >> 1) * Np + (threadIdx.x & 1)] = sum;
        return;
    }
    if (to_image) {
        // the diagonal tile of the step it is factored in: straight into the LDS image diag_body works on (lower
        // triangle, zeros above) instead of out to memory and back
        __syncthreads();                                           // the staging buffers the image overlays are drained
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int r = wr * 64 + tm * 32 + accrow(q, lane);
                    const int c = wc * 64 + tn * 32 + (lane & 31);
                    smem[r * (TS + 1) + c] = (c <= r) ? -acc[tm * 2 + tn][q] : 0.f;
                }
        return;
    }
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
}

// ----------------------------------------------------------------------------- P2
// One workgroup per matrix factors the 128x128 diagonal block and inverts it.  128 dependent pivots
// make this a latency chain, so the block lives in REGISTERS: 16x16 threads, thread (ty,tx) owns the
// 8x8 elements (ty+16*ii, tx+16*cc) (cyclic, so every thread stays busy as the active window
// shrinks).  Per pivot: the 16 owners of column j publish it to a 512-byte LDS buffer (double
// buffered -> one barrier per pivot), every thread reads its 8 row- and 8 column-entries with four
// ds_read_b128 and applies the rank-1 update to the registers that are still active; which (ii,cc)
// pairs are active is decided at compile time (the 16-pivot groups are unrolled), only the
// group's own block row/column needs a lane mask.  The finished columns are also written to a row-major
// LDS image of L for the second phase, the inverse W = L^-1, which is blocked by 32 and runs on the
// matrix cores (see below).  Broadcast-vector element order: index (i%16)*8 + i/16, so a thread's 8
// entries are contiguous.
constexpr int DT = TS + 1;                                      // row stride of the tile image: strided b32 reads conflict-free
constexpr int DIAG_LDS_FLOATS = TS * DT;                        // 66,048 B, fits the GEMM staging area

// 32x32x32 products on fp32 MFMA for the blocked inverse below.  A (and B) are 32x32 blocks of the LDS tile image
// (row stride DT); "reg" variants take the B operand straight from an accumulator: register q of lane (c, h) holds
// B[p_q + 4h][c], p_q = (q&3) + 8(q>>2), which is exactly what MFMA step q wants if A supplies column p_q + 4h.
// Operands are fetched for the whole product first (32 independent LDS reads in flight) and the 16 MFMAs then issue
// back to back: left interleaved, every step waited for its own two reads (1.0 us per product instead of 0.45).
__device__ __forceinline__ f32x16 mm32_lds_lds(f32x16 acc, const float* __restrict__ A, const float* __restrict__ B) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    float av[16], bv[16];
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
        const int kk = 2 * s2 + lh;
        av[s2] = A[l31 * DT + kk];
        bv[s2] = B[kk * DT + l31];
    }
    VOLT_SB();
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bv[s2], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ f32x16 mm32_lds_reg(f32x16 acc, const float* __restrict__ A, const f32x16& Breg) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    float av[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) av[q] = A[l31 * DT + (q & 3) + 8 * (q >> 2) + 4 * lh];
    VOLT_SB();
#pragma unroll
    for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], Breg[q], acc, 0, 0, 0);
    return acc;
}

// acc += (neg ? -1 : 1) * A B^T for 32x32 blocks of the LDS image: A[r][p], B[c][p] both row-major (stride DT).
template <bool NEG>
__device__ __forceinline__ f32x16 mm32_nt(f32x16 acc, const float* __restrict__ A, const float* __restrict__ B) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    float av[16], bv[16];
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
        const int kk = 2 * s2 + lh;
        av[s2] = A[l31 * DT + kk];
        bv[s2] = B[l31 * DT + kk];
    }
    VOLT_SB();
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(NEG ? -av[s2] : av[s2], bv[s2], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ float lane_bcast(float v, int src) {       // readlane is an integer builtin: bit-cast
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// acc_i = fma(-b_i, readlane(src_i, lane_i), acc_i): SGPR broadcast + FMA pinned together in one asm block.  Left to
// the compiler the broadcasts (independent of the FMA chains) are all hoisted to the top and spilled lane by lane
// (v_writelane), doubling the instruction count of the pivot loop.  A VALU may read a readlane's SGPR two wait
// states after it: three broadcasts in a row cover that for each other, shorter groups pad with s_nop.
#define VOLT_RL "v_readlane_b32 "
__device__ __forceinline__ void rl_fma3(float& c0, float& c1, float& c2, float b0, float b1, float b2, float s0,
                                        float s1, float s2, int l0, int l1, int l2) {
    float t0, t1, t2;
    asm volatile(VOLT_RL "%3, %9, %12\n\t" VOLT_RL "%4, %10, %13\n\t" VOLT_RL "%5, %11, %14\n\t"
                 "v_fma_f32 %0, -%6, %3, %0\n\tv_fma_f32 %1, -%7, %4, %1\n\tv_fma_f32 %2, -%8, %5, %2"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "=&s"(t0), "=&s"(t1), "=&s"(t2)
                 : "v"(b0), "v"(b1), "v"(b2), "v"(s0), "v"(s1), "v"(s2), "i"(l0), "i"(l1), "i"(l2));
}
__device__ __forceinline__ void rl_fma2(float& c0, float& c1, float b0, float b1, float s0, float s1, int l0, int l1) {
    float t0, t1;
    asm volatile(VOLT_RL "%2, %6, %8\n\t" VOLT_RL "%3, %7, %9\n\ts_nop 0\n\t"
                 "v_fma_f32 %0, -%4, %2, %0\n\tv_fma_f32 %1, -%5, %3, %1"
                 : "+v"(c0), "+v"(c1), "=&s"(t0), "=&s"(t1)
                 : "v"(b0), "v"(b1), "v"(s0), "v"(s1), "i"(l0), "i"(l1));
}
// A VALU result needs one wait state before v_readlane may read that VGPR (the hardware does not interlock this
// path and the compiler cannot see into the asm blocks): tie a one-cycle nop to the value.
__device__ __forceinline__ void settle(float& v) { asm volatile("s_nop 0" : "+v"(v)); }

__device__ __forceinline__ void rl_fma1(float& c0, float b0, float s0, int l0) {
    float t0;
    asm volatile(VOLT_RL "%1, %3, %4\n\ts_nop 1\n\tv_fma_f32 %0, -%2, %1, %0"
                 : "+v"(c0), "=&s"(t0)
                 : "v"(b0), "v"(s0), "i"(l0));
}

// Phase stamps (tuning hook volt_tune_diag_f32 only): s_memtime of thread 0 at the phase boundaries
#define VOLT_STAMP(i)                                                                  \
    do {                                                                               \
        if (STAMP && threadIdx.x == 0) stamps[i] = __builtin_amdgcn_s_memtime();       \
    } while (0)

// The 128x128 diagonal block is factored in four sub-block columns of 32.  A wave keeps one matrix row per lane in
// registers; per pivot the pivot and the column entries travel by v_readlane (SGPR broadcast), so the dependent pivots
// cost no barrier and no LDS round trip -- but every element update is a readlane + FMA pair (5 clocks each, measured:
// scripts/ubench/readlane.hip), and that instruction count is what the block's latency is made of.  Two things keep
// it down:
//  * recursion by 16: only the two 16x16 diagonal quarters of a sub-block go through the pivot loop; the Schur
//    complement between them runs on v_mfma_f32_16x16x4_f32 (through a small private LDS scratch for the layout change);
//  * the rows BELOW the sub-block ride along: lanes 32..63 of a pivot wave carry the 32 rows of one panel block
//    (kb+1+wave, kb), which the very same instructions turn into L[i,kb] = A[i,kb] L_kk^-T.  Waves 0..2 each repeat the
//    (tiny) diagonal factorisation for their own panel block, so NO inverse is needed on the way down: X_kb = L_kk^-1
//    and the rows of W = L^-1 are worked out one phase later by the waves that have no panel rows left.
// pivots j in [J0, J1) of the lane-per-row factorisation, updating columns (j, J1)
template <int J0, int J1>
__device__ __forceinline__ void pivots16(float (&a)[32], float (&rv)[32], int& npos) {
    settle(a[J0]);
    float d = lane_bcast(a[J0], J0);
    float rinv = __builtin_amdgcn_rsqf(d);
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        npos += (d > 0.f) ? 1 : 0;                                          // non-positive and NaN pivots are not counted
        rv[j] = rinv;
        float l = a[j] * rinv;                                              // lane r: L[r][j]; lane j: sqrt(d)
        settle(l);
        a[j] = l;
        // a[c] -= L[r][j] L[c][j] for j < c < J1 (valid where r >= c); the next pivot's column first, so that its
        // broadcast and rsq are in flight under the other columns' updates
        float rnext = 0.f;
        if (j + 1 < J1) {
            rl_fma1(a[j + 1], l, l, j + 1);
            settle(a[j + 1]);
            d = lane_bcast(a[j + 1], j + 1);
            rnext = __builtin_amdgcn_rsqf(d);
        }
        int c = j + 2;
#pragma unroll
        for (; c + 2 < J1; c += 3) rl_fma3(a[c], a[c + 1], a[c + 2], l, l, l, l, l, l, c, c + 1, c + 2);
        if (c + 1 < J1) rl_fma2(a[c], a[c + 1], l, l, l, l, c, c + 1);
        else if (c < J1) rl_fma1(a[c], l, l, c);
        rinv = rnext;
    }
}
// rows [R0, R1) of the inverse of the 16x16 diagonal quarter starting at R0: lane c solves L x = e_c, column
// oriented so that the FMAs of one step are independent: x[m] = acc[m] / L[m][m], then acc[r] -= L[r][m] x[m] for
// r > m, with L[r][m] broadcast from lane r's register a[m]
template <int R0, int R1>
__device__ __forceinline__ void invert16(const float (&a)[32], const float (&rv)[32], float (&x)[32], int l31) {
#pragma unroll
    for (int r = R0; r < R1; ++r) x[r] = (r == l31) ? 1.f : 0.f;
#pragma unroll
    for (int m = R0; m < R1; ++m) {
        const float xm = x[m] * rv[m];
        x[m] = xm;
        int r = m + 1;
#pragma unroll
        for (; r + 2 < R1; r += 3) rl_fma3(x[r], x[r + 1], x[r + 2], xm, xm, xm, a[m], a[m], a[m], r, r + 1, r + 2);
        if (r + 1 < R1) rl_fma2(x[r], x[r + 1], xm, xm, a[m], a[m], r, r + 1);
        else if (r < R1) rl_fma1(x[r], xm, a[m], r);
    }
}
// LDS hand-over between the lanes of ONE wave (its LDS operations execute in order): keep the compiler from moving
// accesses across, no instruction is needed
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int PSC = 17;                                         // row stride of the 16x16 private scratch tiles
constexpr int PIVOT_SCRATCH_FLOATS = 2 * 16 * PSC;              // per pivot wave: L21 and the updated A22
static_assert(DIAG_LDS_FLOATS + 3 * PIVOT_SCRATCH_FLOATS <= 2 * STAGE_FLOATS, "pivot scratch must fit behind the image");

// One pivot wave, sub-block column kb: lanes 0..31 hold the rows of the diagonal sub-block (kb,kb), lanes 32..63 the
// rows of the panel block (prow,kb) -- or mirror lanes 0..31 when prow < 0.  The diagonal sub-block in the image is
// only READ here (several waves factor it side by side); the panel block is rewritten in place with L[prow,kb]; the
// factored diagonal rows come back in a[] (lane r < 32: L_kk[r][0..r]).
template <bool STAMP = false>
__device__ __forceinline__ void pivot_phase(float* __restrict__ sT, int kb, int prow, float (&a)[32], int& npos,
                                            long long* stamps = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, l15 = lane & 15, g = lane >> 4;
    const bool panel = prow >= 0;                                           // wave-uniform
    const bool prw = panel && lane >= 32;                                   // this lane carries a panel row
    const bool low = !prw && l31 >= 16;                                     // ... a row of the lower diagonal half
    float* Dk = sT + (32 * kb) * DT + 32 * kb;
    float* Rw = prw ? sT + (32 * prow + l31) * DT + 32 * kb : Dk + l31 * DT;
    float* S21 = sT + DIAG_LDS_FLOATS + wave * PIVOT_SCRATCH_FLOATS;        // L21[16][16] of this wave's copy
    float* S22 = S21 + 16 * PSC;                                            // A22 - L21 L21^T
    float rv[32];
    // ---- left half: L11 (lanes 0..15), L21 = A21 L11^-T (lanes 16..31), P1 = A[prow,kb][:, :16] L11^-T (lanes 32..63)
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = Rw[c];
    if (STAMP && kb == 1) VOLT_STAMP(16);
    pivots16<0, 16>(a, rv, npos);
    if (STAMP && kb == 1) VOLT_STAMP(17);
    {
        float* dst = prw ? Rw : S21 + (l31 & 15) * PSC;
        if (prw || low) {
#pragma unroll
            for (int c = 0; c < 16; ++c) dst[c] = a[c];
        }
    }
    wave_lds_fence();
    // ---- Schur complement on the matrix cores: [A22; P2] -= [L21; P1] L21^T, B[k][j] = L21[j][k] = L21[l15][4s + g]
    float v21[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) v21[s] = S21[l15 * PSC + 4 * s + g];
    {
        f32x4 c22;
#pragma unroll
        for (int q = 0; q < 4; ++q) c22[q] = Dk[(16 + 4 * g + q) * DT + 16 + l15];
#pragma unroll
        for (int s = 0; s < 4; ++s) c22 = __builtin_amdgcn_mfma_f32_16x16x4f32(-v21[s], v21[s], c22, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) S22[(4 * g + q) * PSC + l15] = c22[q];
    }
    if (panel) {
        float* Pk = sT + (32 * prow) * DT + 32 * kb;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 cp;
#pragma unroll
            for (int q = 0; q < 4; ++q) cp[q] = Pk[(16 * t + 4 * g + q) * DT + 16 + l15];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                cp = __builtin_amdgcn_mfma_f32_16x16x4f32(-Pk[(16 * t + l15) * DT + 4 * s + g], v21[s], cp, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) Pk[(16 * t + 4 * g + q) * DT + 16 + l15] = cp[q];
        }
    }
    wave_lds_fence();
    if (STAMP && kb == 1) VOLT_STAMP(18);
    // ---- right half: L22 in lanes 16..31 (lanes 0..15 carry zeros), P2 L22^-T in lanes 32..63
    {
        const float* src = prw ? Rw + 16 : S22 + (l31 & 15) * PSC;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float t = src[c];
            a[16 + c] = (prw || low) ? t : 0.f;
        }
    }
    if (STAMP && kb == 1) VOLT_STAMP(19);
    pivots16<16, 32>(a, rv, npos);
    if (STAMP && kb == 1) VOLT_STAMP(20);
    if (prw) {
#pragma unroll
        for (int c = 16; c < 32; ++c) Rw[c] = a[c];
    }
}

// Stores into the 128x128 block W_k go through one buffer descriptor: lane part of the address in ONE 32-bit VGPR,
// the row part as a constant scalar offset.  (As flat stores the rows are 512 B apart, beyond the 12-bit immediate:
// the compiler then keeps a 64-bit address pair per store and hoists them all -- 190 spilled VGPRs.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t w_rsrc(float* W) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, TS * TS * 4, 0x00020000);
}
// AUX = 16 (sc1): written through at agent scope -- what a workgroup on another XCD reads with an sc1 load once it has
// seen the flag, with no L2-wide write-back / invalidate on either side (small_step_kernel's slab hand-off).
constexpr int AUX_SC1 = 16;
template <int AUX = 0>
__device__ __forceinline__ void w_store(__amdgpu_buffer_rsrc_t rs, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, AUX);
}
// One wave: X = L_kk^-1 in place of L_kk in the image, and out to the diagonal block kb of W in memory -- all but
// W[0][0], the ready flag, which is published last.  The two 16x16 diagonal quarters are inverted SIDE BY SIDE, X11 in
// lanes 0..15 and X22 in lanes 32..47 (lane c solves L x = e_c by forward substitution); the L entries are the same
// for every lane of a half-wave and come as LDS broadcast reads -- one ds_read + one FMA per term and no row
// registers, where the pivot loop's readlane idiom would need two instructions and serve one quarter at a time.  The
// off-diagonal quarter X21 = -X22 L21 X11 runs on the matrix cores.
template <bool SLABS = false>
__device__ __forceinline__ void x_block(float* __restrict__ sT, int kb, __amdgpu_buffer_rsrc_t rs) {
    constexpr int AUX = SLABS ? AUX_SC1 : 0;
    constexpr bool store00 = SLABS;
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4, h = lane >> 5;
    const bool up = (lane & 16) != 0;                                       // lanes 16..31, 48..63: no column of their own
    float* Dk = sT + (32 * kb) * DT + 32 * kb;
    const float* Lh = Dk + (16 * h) * DT + 16 * h;                          // this half-wave's diagonal quarter
    float v21[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) v21[s] = Dk[(16 + l15) * DT + 4 * s + g];   // L21 in MFMA operand layout
    float x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float acc = (r == l15) ? 1.f : 0.f;
#pragma unroll
        for (int m = 0; m < r; ++m) acc = __builtin_fmaf(-Lh[r * DT + m], x[m], acc);
        const float d = Lh[r * DT + r];
        const float ri = __builtin_amdgcn_rcpf(d);
        x[r] = acc * __builtin_fmaf(__builtin_fmaf(-d, ri, 1.f), ri, ri);   // one Newton step on the hardware reciprocal
    }
    wave_lds_fence();
    if (!up) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Dk[(16 * h + r) * DT + 16 * h + l15] = x[r];      // X11, X22 (zeros above stay)
    }
    wave_lds_fence();
    // T = L21 X11 lands as T[4g + q][l15] in register q, which is the B operand of a product whose k index runs
    // 4g + q at step q; A follows that order
    f32x4 t21 = {0.f, 0.f, 0.f, 0.f}, r21 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s)
        t21 = __builtin_amdgcn_mfma_f32_16x16x4f32(v21[s], Dk[(4 * s + g) * DT + l15], t21, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        r21 = __builtin_amdgcn_mfma_f32_16x16x4f32(Dk[(16 + l15) * DT + 16 + 4 * g + q], t21[q], r21, 0, 0, 0);
    const int wbase = (32 * kb * TS + 32 * kb) * 4;                         // block (kb,kb) of W, bytes
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        Dk[(16 + 4 * g + q) * DT + l15] = -r21[q];
        w_store<AUX>(rs, wbase + ((16 + 4 * g) * TS + l15) * 4, q * TS * 4, -r21[q]);              // X21
    }
    // X11 and X22 from their lanes; lanes 16..31 write the zero quarter above X22
    if (!up || h == 0) {
        const int voff = wbase + (up ? 16 + l15 : (16 * h) * TS + 16 * h + l15) * 4;
        if (kb != 0 || lane != 0 || store00) w_store<AUX>(rs, voff, 0, up ? 0.f : x[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) w_store<AUX>(rs, voff, r * TS * 4, up ? 0.f : x[r]);
    }
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int q = 0; q < 16; ++q) z[q] = 0.f;
    return z;
}
// a finished block (i,j) of W: -R (accumulator layout) to memory and, when later blocks need it as an operand, to an
// image block
__device__ __forceinline__ void w_out(__amdgpu_buffer_rsrc_t rs, int i, int j, float* __restrict__ blk, const f32x16& R) {
    const int lane = threadIdx.x & 63;
    const int voff = ((32 * i + 4 * (lane >> 5)) * TS + 32 * j + (lane & 31)) * 4;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        w_store(rs, voff, ((q & 3) + 8 * (q >> 2)) * TS * 4, -R[q]);
        if (blk) blk[accrow(q, lane) * DT + (lane & 31)] = -R[q];
    }
}

// One workgroup per matrix factors the 128x128 diagonal block D = L L^T and inverts it, W = L^-1.  Phases between
// barriers (kb = sub-block column; "pivot k" = pivot_phase, "X k" = x_block, "W ij" = one 32x32 block of the inverse):
//   A0  waves 0,1,2: pivot 0 with panel blocks (1,0) (2,0) (3,0);  wave 3: the zero blocks of W
//   B0  A[i,1] -= L[i,0] L[1,0]^T, i = 1..3, one block per wave      (wave 0 first parks L_00 in the image and in memory)
//   A1  waves 0,1: pivot 1 with (2,1) (3,1);  wave 2: X 0;  wave 3: step 0's updates of (2,2) (3,2) (3,3)
//   B1  A[i,2] -= L[i,1] L[2,1]^T, i = 2, 3
//   A2  wave 0: pivot 2 with (3,2);  wave 1: X 1;  wave 3: step 1's update of (3,3)
//   B2  A[3,3] -= L[3,2] L[3,2]^T
//   A3  wave 0: pivot 3;  wave 1: X 2;  wave 2: W 10
//   A4  wave 0: X 3;  wave 1: W 20;  wave 2: W 21, then L31 X1;  wave 3: L30 X0 + L31 W10
//   T   wave 3: W 30;  wave 2: W 31;  wave 1: W 32
// W[i,j] = -X_i sum_{m=j}^{i-1} L[i,m] W[m,j] (W[j,j] = X_j).  Every block of W goes to memory from the registers of
// the wave that made it; W10, W20, W21 are also parked in the image blocks (0,1) (0,2) (1,2) ABOVE the diagonal, which
// nothing else uses, as operands for the rows below.  The L blocks stay intact and go out after W_k has been published.
// SLABS (small_step_kernel): the 32-wide column slabs of the block are handed on as they are finished -- the wave that
// inverts sub-block kb-1 also copies the blocks below it, L[kb.., kb-1], out of the image and publishes slab[kb-1] =
// ready_val behind its own release, so that the tiles below this block are solved by substitution (substitute_tile)
// while the later pivots are still running, and nothing on the way down waits for the inverse.
// LOCALPUB (batch_step.hip, batch a multiple of 8): the readers of W_k sit on this XCD -- the block is handed on through its
// L2, behind the storing waves' drain alone: no L2-wide write-back (1.7 - 6.5 us on every block column's critical path).
template <bool STAMP = false, bool SLABS = false, bool LOCALPUB = false>
__device__ __forceinline__ void diag_body(float* __restrict__ A, float* __restrict__ Winv, int* __restrict__ info,
                                          int Np, int k, int b, float* smem, long long* stamps = nullptr,
                                          bool loaded = false, int* ready = nullptr, int ready_val = 0,
                                          int* slab = nullptr, int* pre = nullptr) {
    float* sT = smem;                                    // row-major image (row stride DT): A -> L / X / W
    VOLT_STAMP(0);
    const int n = Np / TS;
    float* D = A + (int64_t)b * Np * Np + (int64_t)k * TS * Np + (int64_t)k * TS;
    float* W = Winv + ((int64_t)b * n + k) * TS * TS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31;

    if (!loaded) {                                       // lower triangle in (unless update_body left it in the image)
        f32x4 v[TS * TS / 4 / NT];
#pragma unroll
        for (int it = 0; it < TS * TS / 4 / NT; ++it) {
            const int e = tid + it * NT, r = e >> 5, c = (e & 31) * 4;
            v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c <= r) v[it] = *reinterpret_cast<const f32x4*>(D + (int64_t)r * Np + c);
        }
#pragma unroll
        for (int it = 0; it < TS * TS / 4 / NT; ++it) {
            const int e = tid + it * NT, r = e >> 5, c = (e & 31) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) sT[r * DT + c + q] = (c + q <= r) ? v[it][q] : 0.f;
        }
    }
    __syncthreads();
    VOLT_STAMP(1);

    auto blk = [&](int i, int j) { return sT + (32 * i) * DT + 32 * j; };
    const __amdgpu_buffer_rsrc_t wrs = w_rsrc(W);
    // A[i,j] -= L[i,m] L[j,m]^T on this wave
    auto trail = [&](int i, int j, int m) {
        float* C = blk(i, j);
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = C[accrow(q, lane) * DT + l31];
        acc = mm32_nt<true>(acc, blk(i, m), blk(j, m));
#pragma unroll
        for (int q = 0; q < 16; ++q) C[accrow(q, lane) * DT + l31] = acc[q];
    };
    // L_kk out of wave 0's registers: to memory, and into the image for x_block
    auto park_l = [&](int kb, const float (&a)[32]) {
        if (lane < 32) {
            float* Dgk = D + (int64_t)(32 * kb + l31) * Np + 32 * kb;
            float* Dk = blk(kb, kb) + l31 * DT;
#pragma unroll
            for (int c4 = 0; c4 < 8; ++c4) {
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = (4 * c4 + q <= l31) ? a[4 * c4 + q] : 0.f;
                    Dk[4 * c4 + q] = v[q];
                }
                *reinterpret_cast<f32x4*>(Dgk + 4 * c4) = v;
            }
        }
    };
    int bad = 0;
#pragma unroll 1
    for (int kb = 0; kb <= 4; ++kb) {
        float a[32];                                     // wave 0: the factored rows of sub-block kb, phase A -> B
        f32x16 P = zero16();                             // kb = 4, waves 2, 3: partial sums of W's last block row, A4 -> T
        // ---- phase A
        const int npw = kb > 3 ? 0 : (kb == 3 ? 1 : 3 - kb);               // pivot waves
        const int xw = kb == 1 ? 2 : (kb == 4 ? 0 : 1);                     // the wave that inverts sub-block kb - 1
        if (wave < npw) {
            int npos = 0;
            pivot_phase<STAMP>(sT, kb, kb < 3 ? kb + 1 + wave : -1, a, npos, stamps);
            if (npos != 32 && bad == 0) {                                   // rare: find the first failed pivot, whose
#pragma unroll                                                              // L[j][j] = d rsq(d) is NaN (d <= 0 or NaN)
                for (int j = 0; j < 32; ++j) {
                    const float ljj = lane_bcast(a[j], j);
                    if (!(ljj > 0.f) && bad == 0) bad = 32 * kb + j + 1;
                }
            }
            if (kb == 3) park_l(3, a);                                      // no other wave reads (3,3) in this phase
        } else if (kb >= 1 && wave == xw) {
            x_block<SLABS>(sT, kb - 1, wrs);
            if (SLABS) {
                // the blocks below sub-block c out of the image, written through (sc1) like X_c above; the flag follows
                // this wave's own drain -- no L2-wide write-back: the readers use sc1 loads (substitute_tile)
                const int c = kb - 1, row = lane >> 1, c0 = 16 * (lane & 1);
                const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)D, 0, 0x7fffffff, 0x00020000);
                for (int i = c + 1; i <= 3; ++i) {
                    const float* src = blk(i, c) + row * DT + c0;
                    const int voff = (int)((((int64_t)(32 * i + row)) * Np + 32 * c + c0) * 4);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 v = {src[4 * q4], src[4 * q4 + 1], src[4 * q4 + 2], src[4 * q4 + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), drs, voff, 16 * q4, AUX_SC1);
                    }
                }
                // (deferring this drain + flag to the wave's next phase, out of the way of the phase's barrier, was measured:
                // 0.5 - 1 % slower at every size from 1 x 399 to 1 x 4096)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(slab + c, ready_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (kb == 0) {                                               // wave 3: the zero blocks of W above the diagonal
            if (SLABS && pre && lane == 0) {            // and the hand-on of the tile the caller finished before this block:
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // every wave drained its stores ahead of the
                __hip_atomic_store(pre, ready_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // barrier above
            }
            const u32x4 z = {0u, 0u, 0u, 0u};
            const int voff = ((lane >> 3) * TS + 4 * (lane & 7)) * 4;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = i + 1; j <= 3; ++j)
#pragma unroll
                    for (int it = 0; it < 4; ++it)
                        __builtin_amdgcn_raw_buffer_store_b128(z, wrs, voff, ((32 * i + 8 * it) * TS + 32 * j) * 4, 0);
        } else if (wave == 3 && kb <= 2) {                                  // what step kb-1 still owes the blocks right of column kb
            for (int i = kb + 1; i <= 3; ++i)
                for (int j = kb + 1; j <= i; ++j) trail(i, j, kb - 1);
        } else if (kb == 3 && wave == 2) {
            f32x16 S = mm32_lds_lds(zero16(), blk(1, 0), blk(0, 0));                                  // L10 X0
            w_out(wrs, 1, 0, blk(0, 1), mm32_lds_reg(zero16(), blk(1, 1), S));                      // W10 = -X1 S
        } else if (kb == 4 && wave == 1) {
            f32x16 S = mm32_lds_lds(zero16(), blk(2, 0), blk(0, 0));                                  // L20 X0
            S = mm32_lds_lds(S, blk(2, 1), blk(0, 1));                                                // + L21 W10
            w_out(wrs, 2, 0, blk(0, 2), mm32_lds_reg(zero16(), blk(2, 2), S));                      // W20 = -X2 S
        } else if (kb == 4 && wave == 2) {
            f32x16 S = mm32_lds_lds(zero16(), blk(2, 1), blk(1, 1));                                  // L21 X1
            w_out(wrs, 2, 1, blk(1, 2), mm32_lds_reg(zero16(), blk(2, 2), S));                      // W21 = -X2 S
            P = mm32_lds_lds(P, blk(3, 1), blk(1, 1));                                                // L31 X1
        } else if (kb == 4 && wave == 3) {
            P = mm32_lds_lds(P, blk(3, 0), blk(0, 0));                                                // L30 X0
            P = mm32_lds_lds(P, blk(3, 1), blk(0, 1));                                                // + L31 W10
        }
        if (STAMP && kb == 1 && lane == 0) stamps[21 + wave] = __builtin_amdgcn_s_memtime();
        if (STAMP && kb == 3 && lane == 0) stamps[25 + wave] = __builtin_amdgcn_s_memtime();
        __syncthreads();
        VOLT_STAMP(2 + 2 * kb);
        if (kb == 4) {
            // ---- phase T: the last block row of W
            if (wave == 3) {
                P = mm32_lds_lds(P, blk(3, 2), blk(0, 2));                                            // + L32 W20
                w_out(wrs, 3, 0, nullptr, mm32_lds_reg(zero16(), blk(3, 3), P));                    // W30 = -X3 P
            } else if (wave == 2) {
                P = mm32_lds_lds(P, blk(3, 2), blk(1, 2));                                            // + L32 W21
                w_out(wrs, 3, 1, nullptr, mm32_lds_reg(zero16(), blk(3, 3), P));                    // W31
            } else if (wave == 1) {
                f32x16 S = mm32_lds_lds(zero16(), blk(3, 2), blk(2, 2));                              // L32 X2
                w_out(wrs, 3, 2, nullptr, mm32_lds_reg(zero16(), blk(3, 3), S));                    // W32
            }
        }
        if (kb >= 3) continue;
        // ---- phase B: L_kk out; the trailing updates of block column kb+1, A[i,kb+1] -= L[i,kb] L[kb+1,kb]^T, one per
        // wave (the blocks further right are caught up by wave 3 during the next pivot phase)
        if (wave == 0) park_l(kb, a);
        if (kb + 1 + wave <= 3) trail(kb + 1 + wave, kb + 1, kb);
        __syncthreads();
        VOLT_STAMP(3 + 2 * kb);
    }
    if (tid == 0 && bad) atomicCAS(info + b, 0, k * TS + bad);          // tid 0 sits in wave 0, which tracked the pivots
    VOLT_STAMP(11);
    // Every block of W went to memory from the wave that made it -- all but the first word: W[0][0] = 1 / L[0][0] is
    // never 0 (NaN for a failed pivot), so it doubles as the "W_k is ready" flag the panel tiles of the same launch
    // poll -- published last, behind an agent-scope release.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    VOLT_STAMP(12);
    if (tid == 0) {
        const int bits = __float_as_int(sT[0]);
        if constexpr (LOCALPUB) {
            *reinterpret_cast<volatile int*>(W) = bits ? bits : 0x7fc00000;     // a plain store: the line stays in this L2
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(reinterpret_cast<int*>(W), bits ? bits : 0x7fc00000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // small_step_kernel: the step-numbered flag it waits on instead (W's first word is never cleared there)
        if (ready) __hip_atomic_store(ready, ready_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    VOLT_STAMP(13);
    // off-diagonal L blocks out (the diagonal sub-blocks went out of wave 0's registers), zeros above: behind the
    // publish, only the next launch reads them
    for (int e = tid; e < TS * TS / 4; e += NT) {
        const int r = e >> 5, c = (e & 31) * 4;
        if ((r >> 5) != (c >> 5)) {
            f32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (c < r) ? sT[r * DT + c + q] : 0.f;
            *reinterpret_cast<f32x4*>(D + (int64_t)r * Np + c) = v;
        }
    }
    VOLT_STAMP(14);
}

// ----------------------------------------------------------------------------- trtri
// Y = L^-T (upper, row-major).  Block row i of X = L^-1 is block column i of Y:
//     X[i,j] = -W_i * T ,  T = sum_{m=j}^{i-1} L[i,m] X[m,j]      (j < i),     X[i,i] = W_i
// Phase 1 (MFMA, K = 128 (i-j)):  T[r][c] = sum_m L[i-rows r, m] * Y[j-rows c, m]   -- both K-contiguous.
// Phase 2 (MFMA, K = 128):        Y[j-rows c, i-cols r] = - sum_p T[p][c] * W_i[r][p]
// -- one tri_tile_run (common.h).  grid part: (i+1) * B tiles (j = 0..i; j == i copies W_i^T).
constexpr int WLD = TS + 4;    // 132-float rows: b128 reads of 16 rows land on 16 distinct slots

// Optional reductions fused into the trtri epilogue (the MLL step needs z = Y'r and ||Y||_F^2; doing
// them here saves a full pass over Y): zpart[b][j][i*128 + r] = sum_c Y[j*128+c][i*128+r] rvec[j*128+c],
// frob[b][tile(j,i)] = sum of squares over rows < N.  Deterministic, no atomics.
struct TriReduce {
    const float* rpad;   // [B,Np] residual, zero padded; nullptr = no reductions
    float* zpart;        // [B,n,Np]
    float* frob;         // [B,n(n+1)/2]
    int N;
};

// Output tiles are written once and not read again before the next launch: non-temporal stores keep them from
// displacing the shared operand in L2 (round 2 A/B, git history: reads 1.70 -> 1.67 GB per launch, +0.2 % speed).
#ifndef VOLT_OUT_NT
#define VOLT_OUT_NT 1
#endif
#if VOLT_OUT_NT
#define VOLT_OUT_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define VOLT_OUT_STORE(p, v) (*(p) = (v))
#endif
// Out[c][r] = -O: O[rb] element (row = c_local, col = r_local): lane & 31 = r, registers = c.
__device__ __forceinline__ void tri_store(const f32x16 (&O)[4], float* __restrict__ Out, int64_t ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = wave * 32 + accrow(q, lane);
            const int r = rb * 32 + l31;
            VOLT_OUT_STORE(Out + (int64_t)c * ldo + r, -O[rb][q]);
        }
}

// The same tile through ONE buffer descriptor with a cache policy of the caller's choice (batch_step.hip: AUX_SC1 | nt --
// written through at agent scope, so that a progress word stored behind the storing waves' own drain is all a reader on
// another XCD needs; no L2-wide write-back per tile).  Lane part of the address in one VGPR, row / block part scalar.
template <int AUX>
__device__ __forceinline__ void tri_store_aux(const f32x16 (&O)[4], float* __restrict__ Out, int64_t ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)Out, 0, 0x7fffffff, 0x00020000);
    const int voff = (int)((((int64_t)(wave * 32 + 4 * (lane >> 5))) * ldo + l31) * 4);
    const int ld4 = (int)(ldo * 4);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int q = 0; q < 16; ++q)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-O[rb][q]), rs, voff, ((q & 3) + 8 * (q >> 2)) * ld4 + rb * 128, AUX);
}

// The same tile TRANSPOSED THROUGH LDS into 16-byte stores: the accumulators hold a tile column per lane (64 dword stores
// per lane, 256 per workgroup -- measured 9.5 us to issue beside a co-resident tile's loads); through the staging area
// (free once the pipeline has ended) a wave turns its own 32 rows into whole 512-byte rows per half-wave: 16 stores per
// lane.  A wave reads back only what it wrote, so no barrier -- LDS operations of one wave execute in order.  Uses
// smem[0, TS * WLD).  Returns with every store of this wave DRAINED (vmcnt = 0).
// HAZARD (gfx950, found the hard way): a 16-byte store reads its data registers some time AFTER it is issued when the
// memory pipeline is backed up, and neither the hardware nor the compiler's hazard recognizer keeps a VALU write (or a
// returning LDS read) of those registers behind it -- a first version that rotated three register sets through the loop
// stored the constant of an unrelated `v_or_b32 v10, 30, ..` that the scheduler had placed right behind
// `buffer_store_dwordx4 v[10:13]` (16 wrong elements in one tile of ~1000).  Hence: all sixteen rows are read into registers
// of their own, then the stores go out back to back with nothing in between, and nothing follows before they have drained.
template <int AUX>
__device__ __forceinline__ void tri_store_lds(const f32x16 (&O)[4], float* __restrict__ Out, int64_t ldo, float* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
    float* mine = smem + wave * 32 * WLD;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int q = 0; q < 16; ++q) mine[accrow(q, lane) * WLD + rb * 32 + l31] = -O[rb][q];
    wave_lds_fence();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)Out, 0, 0x7fffffff, 0x00020000);
    const int voff = (int)((((int64_t)(wave * 32 + lh)) * ldo + 4 * l31) * 4);
    const int ld8 = (int)(ldo * 8);                          // two rows on, in bytes
    f32x4 v[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) v[it] = *reinterpret_cast<const f32x4*>(mine + (2 * it + lh) * WLD + 4 * l31);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    VOLT_SB();
#pragma unroll
    for (int it = 0; it < 16; ++it)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[it]), rs, voff, it * ld8, AUX);
    VOLT_SB();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// diagonal tile of row i: Y[i,i] = W_i^T (transposed through LDS so both sides stay coalesced)
__device__ __forceinline__ void trtri_diag_body(const float* __restrict__ Winv, float* __restrict__ Y, int Np, int i,
                                                int b, TriReduce red, float* smem) {
    const int n = Np / TS;
    float* Yb = Y + (int64_t)b * Np * Np;
    const float* W = Winv + ((int64_t)b * n + i) * TS * TS;
    const int tid = threadIdx.x;
    const int tile_id = i * (i + 1) / 2 + i;             // upper-tile enumeration (cb = i, jb = j)
    for (int e = tid; e < TS * TS / 4; e += NT) {
        const int r = e >> 5, c = (e & 31) * 4;
        *reinterpret_cast<f32x4*>(smem + r * WLD + c) = *reinterpret_cast<const f32x4*>(W + r * TS + c);
    }
    __syncthreads();
    float* Yd = Yb + (int64_t)i * TS * Np + (int64_t)i * TS;
    for (int e = tid; e < TS * TS; e += NT) {
        const int c = e >> 7, r = e & 127;          // Y row c, column r
        Yd[(int64_t)c * Np + r] = (r >= c) ? smem[r * WLD + c] : 0.f;
    }
    if (red.rpad) {
        float* sred = smem + TS * WLD;               // 128 floats behind the W image
        float* rv = sred + TS;                       // the residuals of block row i, staged: read from memory inside the loop
        if (tid < TS) rv[tid] = red.rpad[(int64_t)b * Np + i * TS + tid];   // below they were up to 128 round trips per thread
        __syncthreads();                             // (78 us per diagonal tile beside busy neighbours, 64 x 4096)
        float fz = 0.f, ff = 0.f;
        if (tid < TS) {
            for (int c = 0; c <= tid; ++c) {         // column r = tid of Y: entries W[r][c], c <= r
                const float y = smem[tid * WLD + c];
                fz += y * rv[c];
                if (i * TS + c < red.N) ff += y * y;
            }
            red.zpart[((int64_t)b * n + i) * Np + i * TS + tid] = fz;
            sred[tid] = ff;
        }
        __syncthreads();
        if (tid < 64) {
            const float tot = wave_sum_f(sred[tid] + sred[tid + 64]);
            if (tid == 0) red.frob[(int64_t)b * (n * (n + 1) / 2) + tile_id] = tot;
        }
    }
}

// z-partials and Frobenius partial of an off-diagonal trtri tile (i, j) from its product O (Y = -O).  WT: the z-partials are
// stored written through at agent scope (batch_step.hip: a workgroup of the same launch, possibly on another XCD, adds
// them up behind the tile's progress word).
template <bool WT = false>
// rvs (optional): the 128 residuals of block row j already in LDS (batch_step.hip parks them there at the tile's entry) -- beside a
// co-resident tile in its K loop the sixteen gather loads per lane below queue behind that tile's buffer loads for microseconds
__device__ __forceinline__ void trtri_reduce(const f32x16 (&O)[4], int Np, int i, int j, int b, TriReduce red,
                                             float* smem, const float* rvs = nullptr) {
    const int n = Np / TS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int tile_id = i * (i + 1) / 2 + j;
    // rows of this block row are all < N (only block row n-1 is padded, and that is a diagonal tile)
    const float* rv = red.rpad + (int64_t)b * Np + j * TS + wave * 32;
    float rvq[16];
    if (rvs) {
#pragma unroll
        for (int q = 0; q < 16; ++q) rvq[q] = rvs[wave * 32 + accrow(q, lane)];
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) rvq[q] = rv[accrow(q, lane)];
    }
    float* sz = smem;                                 // [4 waves][128]  (tri_tile_run ended with a barrier)
    float ff = 0.f;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        float cz = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float y = -O[rb][q];
            cz += y * rvq[q];
            ff += y * y;
        }
        cz += __shfl_xor(cz, 32);                     // the two lane halves hold different rows of column r
        if (lh == 0) sz[wave * TS + rb * 32 + l31] = cz;
    }
    ff = wave_sum_f(ff);
    if (lane == 0) sz[4 * TS + wave] = ff;
    __syncthreads();
    if (tid < TS) {
        const float zv = (sz[tid] + sz[TS + tid]) + (sz[2 * TS + tid] + sz[3 * TS + tid]);
        float* zp = red.zpart + ((int64_t)b * n + j) * Np + i * TS + tid;
        if (WT) __hip_atomic_store(zp, zv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *zp = zv;
    }
    if (tid == 0)
        red.frob[(int64_t)b * (n * (n + 1) / 2) + tile_id] = (sz[4 * TS] + sz[4 * TS + 1]) + (sz[4 * TS + 2] + sz[4 * TS + 3]);
}

// ----------------------------------------------------------------------------- panel tile: update + solve in one
// L[i,k] = (A[i,k] - sum_{m<k} L[i,m] L[k,m]^T) W_k^T  for i > k, as ONE two-phase tile (round 1 ran the update and
// the solve as two launches per block column, with the tile written to and read back from HBM in between):
//     T[p][c] = -A[i,k][c][p] + sum_m L[k,m][p] L[i,m][c]          (the NEGATED, TRANSPOSED updated tile)
//     L[i,k][c][r] = -sum_p T[p][c] W_k[r][p]
// W_k is produced by the diagonal workgroup of the SAME launch; it is needed only after the long first phase, so the
// wait (W_k's first word, see diag_body) is normally over before it starts.
// (the tile is set up in factor_step_kernel: panel and trtri tiles share ONE instance of the two-phase pipeline)

// ----------------------------------------------------------------------------- step kernel
// ONE launch per block column k carries everything that is ready at that point, in dispatch order:
//   [0, B)                          the diagonal tile (k,k): apply block m = k-1 (the rest was done one launch earlier by
//                                   the look-ahead), factor, invert -> W_k, publish
//   [B, B + npre)                   diagonal look-ahead: A[k+1,k+1] -= sum_{m<k} L[k+1,m] L[k+1,m]^T (does not need
//                                   column k), so that the next launch's diagonal workgroup starts its 128-pivot chain
//                                   almost at once
//   next (n-k-1) B                  panel tiles (i,k), i > k: update + solve (wait for W_k after their long phase)
//   next k B                        tiles of trtri row k-1 (independent of column k)
// so every launch has ~n*B tiles of comparable length and the diagonal latency chain (46 us) runs beside them.
//   k_upd < 0: no factorisation part (the trailing trtri row, volt_trtri_f32);   i_tri < 0: no trtri part
// What a two-phase tile needs besides its TriTile: where T0 comes from and where the product goes.
struct TriJob {
    TriTile t;
    const float* c0;       // nullptr: T0 = 0 (trtri);  else row `gi` of the input tile, first column of the tile, + 4 (lane >> 5)
    bool row_ok, vec_ok;   // the row exists in the input (not padding) / 16-byte loads are legal
    float* out;            // Out[c * Np + r] = -O
    int i, j, b;           // trtri: tile (i, j) of matrix b (for the reductions);  panel: i = -1
};

template <bool FROMK>
__device__ __forceinline__ TriJob panel_job(float* __restrict__ A, const float* __restrict__ Winv, int Np, int i, int k,
                                            int b, const KSource& src) {
    const int n = Np / TS;
    float* Ab = A + (int64_t)b * Np * Np;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
    const bool usek = FROMK && src.K != nullptr;
    TriJob jb;
    jb.t.X = Ab + (int64_t)k * TS * Np;                   // L[k, 0 ...]   (rows p = columns of the tile)
    jb.t.ldx = Np;
    jb.t.Z = Ab + (int64_t)i * TS * Np;                   // L[i, 0 ...]   (rows c = rows of the tile)
    jb.t.ldz = Np;
    jb.t.n1 = k * (TS / BK);
    jb.t.W = Winv + ((int64_t)b * n + k) * TS * TS;
    jb.t.flag = reinterpret_cast<const int*>(jb.t.W);
    jb.t.want = 0;
    // T0[p][c] = -A[i,k][c][p]: lane owns row c = 32 wave + l31 of the tile; registers 4g..4g+3 of T[tm] are the four
    // consecutive columns p = 32 tm + 8 g + 4 (lane >> 5) + (0..3): one 16-byte load each.  Column block k <= n-2 lies
    // wholly inside the matrix; only the rows of the last block row can be padding.
    const int gi = i * TS + wave * 32 + l31;
    const int64_t ld = usek ? src.ldk : (int64_t)Np;
    jb.c0 = (usek ? src.K + (int64_t)b * src.bsk : Ab) + (int64_t)gi * ld + k * TS + 4 * lh;
    jb.row_ok = !usek || gi < src.N;
    jb.vec_ok = !usek || (((src.ldk & 3) == 0) && ((src.bsk & 3) == 0) && (((uintptr_t)src.K & 15) == 0));
    jb.out = Ab + (int64_t)i * TS * Np + (int64_t)k * TS;
    jb.i = -1;
    jb.j = 0;
    jb.b = b;
    return jb;
}

__device__ __forceinline__ TriJob trtri_job(const float* __restrict__ A, const float* __restrict__ Winv,
                                            float* __restrict__ Y, int Np, int i, int j, int b) {
    const int n = Np / TS;
    const float* Ab = A + (int64_t)b * Np * Np;
    float* Yb = Y + (int64_t)b * Np * Np;
    TriJob jb;
    jb.t.X = Ab + (int64_t)i * TS * Np + (int64_t)j * TS;   // L[i, j*128 ...]
    jb.t.ldx = Np;
    jb.t.Z = Yb + (int64_t)j * TS * Np + (int64_t)j * TS;   // Y[j, j*128 ...]
    jb.t.ldz = Np;
    jb.t.n1 = (i - j) * (TS / BK);
    jb.t.W = Winv + ((int64_t)b * n + i) * TS * TS;
    jb.t.flag = nullptr;
    jb.t.want = 0;
    jb.c0 = nullptr;
    jb.row_ok = jb.vec_ok = true;
    jb.out = Yb + (int64_t)j * TS * Np + (int64_t)i * TS;
    jb.i = i;
    jb.j = j;
    jb.b = b;
    return jb;
}

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
// T0 of a two-phase tile (factor_step_kernel's prologue)
__device__ __forceinline__ void job_t0(const TriJob& jb, f32x16 (&T)[4]) {
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
}

}  // namespace volt
