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
#include "../../include/volt_hip.h"
#include <mutex>
#include <stdlib.h>
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

// ----------------------------------------------------------------------------- P1
// grid.x = (n-k) * B.  Tile t: rows of block (k+t), columns of block k.
// The C tile is loaded into registers BEFORE the K loop (its 64 KB per workgroup would otherwise
// be an un-overlapped read-modify-write at the end: measured 51 -> 83 TF/s at k = 1, 115 -> 125 at
// k = 16) and only stored in the epilogue.  FROMK: the C tile comes straight from the caller's
// K (+ sigma2/jitter on the diagonal, identity in the padding) instead of a prepared copy in A, which
// removes the K -> A copy pass for every block column but the first.
struct KSource {
    const float* K;
    int64_t ldk, bsk;
    const float* sigma2;
    float jitter;
    int N;
};

// Tile (rowblk, colblk) of the working matrix:  C <- C0 - sum_{m = kb0}^{kb1-1} L[rowblk, m] L[colblk, m]^T,
// where C0 is the tile as it stands in A, or -- when `fromk` -- the caller's K (+ sigma2/jitter on the
// diagonal, identity in the padding).  kb0 > 0 continues an update begun by an earlier launch (diagonal
// look-ahead, see factor_step_kernel).
template <bool FROMK>
__device__ __forceinline__ void update_body(float* __restrict__ A, int Np, int rowblk, int colblk, int kb0, int kb1,
                                            bool fromk, int b, const KSource& src, float* smem) {
    float* Ab = A + (int64_t)b * Np * Np;
    const float* Arows = Ab + (int64_t)rowblk * TS * Np + (int64_t)kb0 * TS;   // L[rowblk, kb0:kb1]
    const float* Brows = Ab + (int64_t)colblk * TS * Np + (int64_t)kb0 * TS;   // L[colblk, kb0:kb1]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    float* C = Ab + (int64_t)rowblk * TS * Np + (int64_t)colblk * TS;
    f32x16 acc[4], cpre[4];
    zero_acc(acc);
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
                if (usek) {
                    const int gi = rowblk * TS + r, gj = colblk * TS + c;
                    float v = (gi < src.N && gj < src.N) ? Kb[(int64_t)gi * src.ldk + gj] : 0.f;
                    if (gi == gj) v = (gi < src.N) ? v + add : 1.f;
                    cpre[tm * 2 + tn][q] = v;
                } else {
                    cpre[tm * 2 + tn][q] = C[(int64_t)r * Np + c];
                }
            }
    gemm_nt_128<0>(Arows, Np, Brows, Np, (kb1 - kb0) * (TS / BK), acc, smem);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r = wr * 64 + tm * 32 + accrow(q, lane);
                const int c = wc * 64 + tn * 32 + (lane & 31);
                C[(int64_t)r * Np + c] = cpre[tm * 2 + tn][q] - acc[tm * 2 + tn][q];
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
__device__ __forceinline__ f32x16 mm32_lds_lds(f32x16 acc, const float* __restrict__ A, const float* __restrict__ B) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
        const int kk = 2 * s2 + lh;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * DT + kk], B[kk * DT + l31], acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ f32x16 mm32_lds_reg(f32x16 acc, const float* __restrict__ A, const f32x16& Breg) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int p = (q & 3) + 8 * (q >> 2) + 4 * lh;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * DT + p], Breg[q], acc, 0, 0, 0);
    }
    return acc;
}

// acc += (neg ? -1 : 1) * A B^T for 32x32 blocks of the LDS image: A[r][p], B[c][p] both row-major (stride DT).
template <bool NEG>
__device__ __forceinline__ f32x16 mm32_nt(f32x16 acc, const float* __restrict__ A, const float* __restrict__ B) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
        const int kk = 2 * s2 + lh;
        const float av = A[l31 * DT + kk];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(NEG ? -av : av, B[l31 * DT + kk], acc, 0, 0, 0);
    }
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

// the same with one accumulator: acc -= b0 rl(s0) + b1 rl(s1) + b2 rl(s2)
__device__ __forceinline__ void rl_dot3(float& acc, float b0, float b1, float b2, float s0, float s1, float s2, int ln) {
    float t0, t1, t2;
    asm volatile(VOLT_RL "%1, %7, %10\n\t" VOLT_RL "%2, %8, %10\n\t" VOLT_RL "%3, %9, %10\n\t"
                 "v_fma_f32 %0, -%4, %1, %0\n\tv_fma_f32 %0, -%5, %2, %0\n\tv_fma_f32 %0, -%6, %3, %0"
                 : "+v"(acc), "=&s"(t0), "=&s"(t1), "=&s"(t2)
                 : "v"(b0), "v"(b1), "v"(b2), "v"(s0), "v"(s1), "v"(s2), "i"(ln));
}
__device__ __forceinline__ void rl_dot2(float& acc, float b0, float b1, float s0, float s1, int ln) {
    float t0, t1;
    asm volatile(VOLT_RL "%1, %5, %7\n\t" VOLT_RL "%2, %6, %7\n\ts_nop 0\n\t"
                 "v_fma_f32 %0, -%3, %1, %0\n\tv_fma_f32 %0, -%4, %2, %0"
                 : "+v"(acc), "=&s"(t0), "=&s"(t1)
                 : "v"(b0), "v"(b1), "v"(s0), "v"(s1), "i"(ln));
}
__device__ __forceinline__ void rl_fma1(float& c0, float b0, float s0, int l0) {
    float t0;
    asm volatile(VOLT_RL "%1, %3, %4\n\ts_nop 1\n\tv_fma_f32 %0, -%2, %1, %0"
                 : "+v"(c0), "=&s"(t0)
                 : "v"(b0), "v"(s0), "i"(l0));
}

// One wave factors the 32x32 diagonal sub-block kb of the image and inverts it.  Lane r (both half-waves hold the
// same data) keeps row r in registers; per pivot the pivot and the column entries travel by v_readlane (SGPR
// broadcast), so the 32 dependent pivots cost no barrier and no LDS round trip.  L_kk goes straight to global
// memory; its inverse X = L_kk^-1 (column c solved in lane c, L entries again by readlane) replaces it in the
// image: the panel solve, the trailing updates and the blocked inverse only ever need X.
__device__ __forceinline__ void factor32(float* __restrict__ sT, int kb, float* __restrict__ Dg, int Np, int& bad) {
    const int lane = threadIdx.x & 63, l31 = lane & 31;
    float* Dk = sT + (32 * kb) * DT + 32 * kb;
    float a[32], rv[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) a[c] = Dk[l31 * DT + c];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        settle(a[j]);                                                       // last written inside an asm block
        const float d = lane_bcast(a[j], j);
        if (!(d > 0.f) && bad == 0) bad = 32 * kb + j + 1;                  // non-positive or NaN pivot (wave-uniform)
        const float rinv = __builtin_amdgcn_rsqf(d);
        rv[j] = rinv;
        float l = a[j] * rinv;                                              // lane r: L[r][j]; lane j: sqrt(d)
        settle(l);
        a[j] = l;
        // a[c] -= L[r][j] L[c][j] for c > j (valid where r >= c)
        int c = j + 1;
#pragma unroll
        for (; c + 2 < 32; c += 3) rl_fma3(a[c], a[c + 1], a[c + 2], l, l, l, l, l, l, c, c + 1, c + 2);
        if (c + 1 < 32) rl_fma2(a[c], a[c + 1], l, l, l, l, c, c + 1);
        else if (c < 32) rl_fma1(a[c], l, l, c);
    }
    // L_kk out (zeros above the diagonal); both half-waves write the same words
    float* Dgk = Dg + (int64_t)(32 * kb + l31) * Np + 32 * kb;
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (4 * c4 + q <= l31) ? a[4 * c4 + q] : 0.f;
        *reinterpret_cast<f32x4*>(Dgk + 4 * c4) = v;
    }
    // X = L_kk^-1: lane c solves L x = e_c;  x[r] = (delta_rc - sum_{m<r} L[r][m] x[m]) / L[r][r], with L[r][m]
    // broadcast from lane r's register a[m]  (x[m] == 0 for m < c by construction)
    float x[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        float acc = (r == l31) ? 1.f : 0.f;
        int m = 0;
#pragma unroll
        for (; m + 2 < r; m += 3) rl_dot3(acc, x[m], x[m + 1], x[m + 2], a[m], a[m + 1], a[m + 2], r);
        if (m + 1 < r) rl_dot2(acc, x[m], x[m + 1], a[m], a[m + 1], r);
        else if (m < r) rl_fma1(acc, x[m], a[m], r);
        x[r] = acc * rv[r];
    }
#pragma unroll
    for (int r = 0; r < 32; ++r) Dk[r * DT + l31] = x[r];                   // X[r][c], zero above the diagonal
}

__device__ __forceinline__ void diag_body(float* __restrict__ A, float* __restrict__ Winv, int* __restrict__ info,
                                          int Np, int k, int b, float* smem) {
    float* sT = smem;                                    // row-major image (row stride DT): A -> L / X -> W
    const int n = Np / TS;
    float* D = A + (int64_t)b * Np * Np + (int64_t)k * TS * Np + (int64_t)k * TS;
    float* W = Winv + ((int64_t)b * n + k) * TS * TS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31;

    for (int e = tid; e < TS * TS / 4; e += NT) {        // lower triangle in, zeros above
        const int r = e >> 5, c = (e & 31) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(D + (int64_t)r * Np + c);
#pragma unroll
        for (int q = 0; q < 4; ++q) sT[r * DT + c + q] = (c + q <= r) ? v[q] : 0.f;
    }
    __syncthreads();

    // ---- L = chol(D), blocked by 32: factor32 on one wave, panel solve and trailing updates on the matrix cores.
    // Wave 0 runs ahead: it updates the next diagonal sub-block first and factors it while waves 1..3 finish the
    // other updates of the step.
    int bad = 0;
    if (wave == 0) factor32(sT, 0, D, Np, bad);
    __syncthreads();
#pragma unroll 1
    for (int kb = 0; kb < 3; ++kb) {
        // panel: L[i,kb] = A[i,kb] X_kb^T, block row i = kb+1+wave
        if (kb + 1 + wave <= 3) {
            const int i = kb + 1 + wave;
            float* P = sT + (32 * i) * DT + 32 * kb;
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            acc = mm32_nt<false>(acc, P, sT + (32 * kb) * DT + 32 * kb);
#pragma unroll
            for (int q = 0; q < 16; ++q) P[accrow(q, lane) * DT + l31] = acc[q];
        }
        __syncthreads();
        // trailing updates A[i,j] -= L[i,kb] L[j,kb]^T, kb < j <= i <= 3: (kb+1,kb+1) on wave 0, the rest dealt to 1..3
        int cnt = 0;
        for (int i = kb + 1; i <= 3; ++i)
            for (int j = kb + 1; j <= i; ++j) {
                const bool first = (i == kb + 1);                              // then j == kb+1 too
                const int owner = first ? 0 : 1 + (cnt++ % 3);
                if (wave == owner) {
                    float* C = sT + (32 * i) * DT + 32 * j;
                    f32x16 acc;
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[q] = C[accrow(q, lane) * DT + l31];
                    acc = mm32_nt<true>(acc, sT + (32 * i) * DT + 32 * kb, sT + (32 * j) * DT + 32 * kb);
#pragma unroll
                    for (int q = 0; q < 16; ++q) C[accrow(q, lane) * DT + l31] = acc[q];
                }
            }
        if (wave == 0) factor32(sT, kb + 1, D, Np, bad);
        __syncthreads();
    }
    // off-diagonal L blocks out (the diagonal sub-blocks went out of factor32's registers), zeros above
    for (int e = tid; e < TS * TS / 4; e += NT) {
        const int r = e >> 5, c = (e & 31) * 4;
        if ((r >> 5) != (c >> 5)) {
            f32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (c < r) ? sT[r * DT + c + q] : 0.f;
            *reinterpret_cast<f32x4*>(D + (int64_t)r * Np + c) = v;
        }
    }

    // ---- W = L^-1, blocked by 32 on the matrix cores: wave j < 3 owns block column j,
    // W[i,j] = -X_i * sum_{m=j}^{i-1} L[i,m] W[m,j], top to bottom; the W[m,j] it produced stay in its accumulators
    // and are fed back as B operands from registers; then the W blocks replace the L blocks and the image goes out.
    f32x16 Wr[3];
    if (wave < 3) {
        const int j = wave;
#pragma unroll
        for (int di = 1; di <= 3; ++di) {
            const int i = j + di;
            if (i <= 3) {                                                     // wave-uniform
                f32x16 S;
#pragma unroll
                for (int q = 0; q < 16; ++q) S[q] = 0.f;
                S = mm32_lds_lds(S, sT + (32 * i) * DT + 32 * j, sT + (32 * j) * DT + 32 * j);       // L[i,j] X_j
#pragma unroll
                for (int dm = 1; dm < di; ++dm)
                    S = mm32_lds_reg(S, sT + (32 * i) * DT + 32 * (j + dm), Wr[dm - 1]);              // L[i,m] W[m,j]
                f32x16 R;
#pragma unroll
                for (int q = 0; q < 16; ++q) R[q] = 0.f;
                R = mm32_lds_reg(R, sT + (32 * i) * DT + 32 * i, S);                                  // X_i S
#pragma unroll
                for (int q = 0; q < 16; ++q) Wr[di - 1][q] = -R[q];
            }
        }
    }
    __syncthreads();                                                          // every L block has been consumed
    if (wave < 3) {
        const int j = wave;
#pragma unroll
        for (int di = 1; di <= 3; ++di) {
            const int i = j + di;
            if (i <= 3) {
#pragma unroll
                for (int q = 0; q < 16; ++q) sT[(32 * i + accrow(q, lane)) * DT + 32 * j + l31] = Wr[di - 1][q];
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < TS * TS / 4; e += NT) {
        const int r = e >> 5, c = (e & 31) * 4;
        f32x4 w4;
#pragma unroll
        for (int q = 0; q < 4; ++q) w4[q] = (c + q <= r) ? sT[r * DT + c + q] : 0.f;
        *reinterpret_cast<f32x4*>(W + r * TS + c) = w4;
    }
    if (tid == 0 && bad) atomicCAS(info + b, 0, k * TS + bad);          // tid 0 sits in wave 0, which tracked the pivots
}

// ----------------------------------------------------------------------------- P3
// grid.x = (n-k-1) * B.  L[i,k] = A[i,k] * W_k^T, in place (the tile is fully staged through LDS
// before the epilogue stores).
__global__ __launch_bounds__(256, 2) void potrf_trsm_kernel(float* __restrict__ A, const float* __restrict__ Winv,
                                                           int Np, int k, int B) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    const int n = Np / TS;
    int t, b;
    decode_tile_batch(n - k - 1, B, t, b);
    float* P = A + (int64_t)b * Np * Np + (int64_t)(k + 1 + t) * TS * Np + (int64_t)k * TS;
    const float* W = Winv + ((int64_t)b * n + k) * TS * TS;
    f32x16 acc[4];
    zero_acc(acc);
    gemm_nt_128<0>(P, Np, W, TS, TS / BK, acc, smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r = wr * 64 + tm * 32 + accrow(q, lane);
                const int c = wc * 64 + tn * 32 + (lane & 31);
                P[(int64_t)r * Np + c] = acc[tm * 2 + tn][q];
            }
}

// ----------------------------------------------------------------------------- trtri
// Y = L^-T (upper, row-major).  Block row i of X = L^-1 is block column i of Y:
//     X[i,j] = -W_i * T ,  T = sum_{m=j}^{i-1} L[i,m] X[m,j]      (j < i),     X[i,i] = W_i
// Phase 1 (MFMA, K = 128 (i-j)):  T[r][c] = sum_m L[i-rows r, m] * Y[j-rows c, m]   -- both K-contiguous.
// Phase 2 (MFMA, K = 128):        Y[j-rows c, i-cols r] = - sum_p T[p][c] * W_i[r][p]
//   With the 1x4 wave layout a wave holds all 128 p for its 32 columns c, and the accumulator
//   layout of T (lane = c, registers = p) is exactly the A-operand layout of phase 2, so T never
//   leaves the register file; W_i is staged once in LDS.
// grid.x = (i+1) * B (tile j = 0..i; j == i copies W_i^T).
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

__device__ __forceinline__ void trtri_body(const float* __restrict__ A, const float* __restrict__ Winv,
                                           float* __restrict__ Y, int Np, int i, int j, int b, TriReduce red,
                                           float* smem) {
    const int n = Np / TS;
    const float* Ab = A + (int64_t)b * Np * Np;
    float* Yb = Y + (int64_t)b * Np * Np;
    const float* W = Winv + ((int64_t)b * n + i) * TS * TS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int tile_id = i * (i + 1) / 2 + j;             // upper-tile enumeration (cb = i, jb = j)

    if (j == i) {
        // diagonal tile: Y[i,i] = W_i^T (transposed through LDS so both sides stay coalesced)
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
            float fz = 0.f, ff = 0.f;
            if (tid < TS) {
                const float* rv = red.rpad + (int64_t)b * Np + i * TS;
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
        return;
    }

    // One software pipeline over n1 + 4 chunks: chunks [0,n1) are phase 1 (A = L rows of block i, B = Y
    // rows of block j, both at K offset 32c), chunks [n1,n1+4) are phase 2, whose B tile is W_i[:, 32tp..]
    // and whose A operand is T itself, straight out of the accumulator registers -- so W_i streams
    // through the same double-buffered LDS stage as everything else and there is no staging bubble
    // between the phases.
    const int n1 = (i - j) * (TS / BK);
    const float* Lrows = Ab + (int64_t)i * TS * Np + (int64_t)j * TS;   // L[i, j*128 ...]
    const float* Yrows = Yb + (int64_t)j * TS * Np + (int64_t)j * TS;   // Y[j, j*128 ...]
    f32x16 T[4], O[4];
    zero_acc(T);
    zero_acc(O);
    // Two register stage sets (loads issued two chunks ahead of their LDS write), buffer-addressed.
    StageRegs sr0, sr1;
    const int srow = tid >> 3, scq = (tid & 7) * 4;
    const StageAddr sa = stage_addr(Lrows, Np, Yrows, Np);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, TS * TS * 4, 0x00020000);
    int vw[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) vw[p] = ((srow + 32 * p) * TS + scq) * 4;
    auto load_chunk = [&](StageRegs& sr, int c) {
        if (c < n1) {
            stage_load_buf(sr, sa, c * BK);
        } else {
            const int so = __builtin_amdgcn_readfirstlane((c - n1) * BK * 4);
#pragma unroll
            for (int p = 0; p < 4; ++p)
                sr.b[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, vw[p], so, 0));
        }
    };
    auto store_chunk = [&](const StageRegs& sr, int c, float* buf) {
        if (c < n1) {
            stage_store(sr, buf);
        } else {
            float* sB = buf + TS * SLD;
#pragma unroll
            for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(sB + (srow + 32 * p) * SLD + scq) = sr.b[p];
        }
    };
    const int nall = n1 + 4;                                // >= 5
    load_chunk(sr0, 0);
    store_chunk(sr0, 0, smem);
    load_chunk(sr0, 1);                                     // even-numbered stores use sr1, odd use sr0
    load_chunk(sr1, 2);
    __syncthreads();
    // chunk c lives in buffer c & 1; at its mid-point chunk c+1 is written (from sr0 if c is even, else
    // sr1) and chunk c+3 is requested into the same set.
#define VOLT_TRI_STAGE(C)                                                            \
    {                                                                                \
        float* nxt_ = smem + (((C) + 1) & 1) * STAGE_FLOATS;                         \
        if (((C) & 1) == 0) {                                                        \
            if ((C) + 1 < nall) store_chunk(sr0, (C) + 1, nxt_);                     \
            if ((C) + 3 < nall) load_chunk(sr0, (C) + 3);                            \
        } else {                                                                     \
            if ((C) + 1 < nall) store_chunk(sr1, (C) + 1, nxt_);                     \
            if ((C) + 3 < nall) load_chunk(sr1, (C) + 3);                            \
        }                                                                            \
    }
    int c = 0;
    for (; c + 1 < n1; c += 2) {                            // phase 1, two chunks per trip (static set names)
        mma_chunk<1, 0, BK / 16>(smem, T);
        VOLT_TRI_STAGE(c)
        mma_chunk<1, BK / 16, BK / 8>(smem, T);
        __syncthreads();
        mma_chunk<1, 0, BK / 16>(smem + STAGE_FLOATS, T);
        VOLT_TRI_STAGE(c + 1)
        mma_chunk<1, BK / 16, BK / 8>(smem + STAGE_FLOATS, T);
        __syncthreads();
    }
    // n1 is a multiple of 4 (128-wide blocks of 32-wide chunks), so phase 1 always ends on an even chunk
    // phase 2: out[cc][r] for cc in this wave's 32 columns, r in 4 blocks of 32.
    // registers 4g..4g+3 of T[tp] <-> p = tp*32 + 8g + 4*lh + (0..3); B fragment = W[r][p] from the stage.
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        const int cc = n1 + tp;                             // parity of cc == parity of tp
        const float* sB = smem + (tp & 1) * STAGE_FLOATS + TS * SLD;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g == 2) {
                float* nxt_ = smem + ((tp + 1) & 1) * STAGE_FLOATS;
                if ((tp & 1) == 0) {
                    if (cc + 1 < nall) store_chunk(sr0, cc + 1, nxt_);
                    if (cc + 3 < nall) load_chunk(sr0, cc + 3);
                } else {
                    if (cc + 1 < nall) store_chunk(sr1, cc + 1, nxt_);
                    if (cc + 3 < nall) load_chunk(sr1, cc + 3);
                }
            }
            // W_i is lower triangular: W[r][p] = 0 for p > r, so row blocks rb < tp contribute nothing
            f32x4 w[4];
#pragma unroll
            for (int rb = tp; rb < 4; ++rb)
                w[rb] = *reinterpret_cast<const f32x4*>(sB + (rb * 32 + l31) * SLD + 8 * g + 4 * lh);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float a = T[tp][4 * g + m];
#pragma unroll
                for (int rb = tp; rb < 4; ++rb)
                    O[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w[rb][m], O[rb], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#undef VOLT_TRI_STAGE
    // O[rb] element (row = c_local, col = r_local): lane&31 = r, registers = c.  Y = -O.
    float* Yt = Yb + (int64_t)j * TS * Np + (int64_t)i * TS;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = wave * 32 + accrow(q, lane);
            const int r = rb * 32 + l31;
            Yt[(int64_t)c * Np + r] = -O[rb][q];
        }
    if (red.rpad) {
        // rows of this block row are all < N (only block row n-1 is padded, and that is a diagonal tile)
        const float* rv = red.rpad + (int64_t)b * Np + j * TS + wave * 32;
        float rvq[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) rvq[q] = rv[accrow(q, lane)];
        __syncthreads();                                  // everyone is done reading W from smem
        float* sz = smem;                                 // [4 waves][128]
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
        if (tid < TS)
            red.zpart[((int64_t)b * n + j) * Np + i * TS + tid] =
                (sz[tid] + sz[TS + tid]) + (sz[2 * TS + tid] + sz[3 * TS + tid]);
        if (tid == 0)
            red.frob[(int64_t)b * (n * (n + 1) / 2) + tile_id] = (sz[4 * TS] + sz[4 * TS + 1]) + (sz[4 * TS + 2] + sz[4 * TS + 3]);
    }
}

// ----------------------------------------------------------------------------- fused step kernel
// One launch per block column k carries everything that is ready at that point:
//   workgroups [0, (n-k)B)          P1 tiles of column k; the workgroup that owns the diagonal tile (t = 0)
//                                   goes straight on to factor and invert it (P2) -- the 46 us latency chain
//                                   runs next to the other tiles instead of after them
//   workgroups [(n-k)B, (n-k)B+kB)  tiles of trtri row k-1 (independent of column k)
// so every launch has n*B tiles of comparable length, longest first (diagonal, P1 tiles with K = 128k,
// trtri tiles with K = 128(k-j)), instead of three launches of (n-k)B, B and kB.
//   k_upd < 0: no P1/P2 part (used for the trailing trtri row and by volt_trtri_f32)
//   i_tri < 0: no trtri part (forward-only factorisation)
template <bool FROMK>
__global__ __launch_bounds__(256, 2) void factor_step_kernel(float* __restrict__ A, float* __restrict__ Winv,
                                                            float* __restrict__ Y, int* __restrict__ info, int Np,
                                                            int k_upd, int i_tri, int B, KSource src, TriReduce red) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    static_assert(DIAG_LDS_FLOATS <= 2 * STAGE_FLOATS, "diag block must fit the staging area");
    static_assert(TS * WLD + TS <= 2 * STAGE_FLOATS, "W image + reduction scratch must fit");
    const int n = Np / TS;
    // Diagonal look-ahead: the long part of the update of diagonal tile (k+1,k+1), blocks m < k, does not
    // need column k, so it is done HERE (one extra workgroup per matrix, first in the grid); the next
    // launch's diagonal workgroup then only applies block m = k (K = 128) before its 128-pivot chain, and is
    // no longer the longest workgroup of its launch.
    const int npre = (k_upd >= 1 && k_upd + 1 < n) ? B : 0;
    const int nupd = (k_upd >= 0) ? (n - k_upd) * B : 0;
    int w = blockIdx.x;
    if (w < npre) {
        update_body<FROMK>(A, Np, k_upd + 1, k_upd + 1, 0, k_upd, true, w, src, smem);
        return;
    }
    w -= npre;
    if (w < nupd) {
        int t, b;
        decode_tile_batch(w, n - k_upd, B, t, b);
        if (k_upd > 0) {
            if (t == 0 && k_upd >= 2) update_body<FROMK>(A, Np, k_upd, k_upd, k_upd - 1, k_upd, false, b, src, smem);
            else update_body<FROMK>(A, Np, k_upd + t, k_upd, 0, k_upd, true, b, src, smem);
        }
        if (t == 0) {
            if (k_upd > 0) {
                __threadfence_block();           // this workgroup's own C-tile stores, re-read below
                __syncthreads();
            }
            diag_body(A, Winv, info, Np, k_upd, b, smem);
        }
        return;
    }
    int j, b;
    decode_tile_batch(w - nupd, i_tri + 1, B, j, b);
    trtri_body(A, Winv, Y, Np, i_tri, j, b, red, smem);
}

// P1 alone (tuning hook: no diagonal factorisation, so it can be replayed on a finished factor)
__global__ __launch_bounds__(256, 2) void tune_update_kernel(float* __restrict__ A, int Np, int k, int B, KSource src) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    int t, b;
    decode_tile_batch(Np / TS - k, B, t, b);
    update_body<false>(A, Np, k + t, k, 0, k, true, b, src, smem);
}

// Block column 0: no panel update, just the diagonal blocks of the prepared column.
__global__ __launch_bounds__(256, 2) void factor_diag0_kernel(float* __restrict__ A, float* __restrict__ Winv,
                                                             int* __restrict__ info, int Np) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    diag_body(A, Winv, info, Np, 0, blockIdx.x, smem);
}

}  // namespace volt

using namespace volt;

// Optional per-launch timing (bench only): every launch is bracketed by two events on the same
// stream; the caller synchronises and sums the intervals per kernel class.
struct LaunchTimer {
    hipStream_t s;
    std::vector<hipEvent_t> ev;
    std::vector<int> cls;
    explicit LaunchTimer(hipStream_t st) : s(st) {}
    void begin(int c) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        (void)hipEventRecord(e, s);
        ev.push_back(e);
        cls.push_back(c);
    }
    void end() {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        (void)hipEventRecord(e, s);
        ev.push_back(e);
    }
    void collect(float* ms_by_class, int* n_by_class, int nclass) {
        (void)hipStreamSynchronize(s);
        for (int c = 0; c < nclass; ++c) { ms_by_class[c] = 0.f; n_by_class[c] = 0; }
        for (size_t i = 0; i < cls.size(); ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
            ms_by_class[cls[i]] += ms;
            n_by_class[cls[i]] += 1;
        }
        for (auto e : ev) (void)hipEventDestroy(e);
        ev.clear();
        cls.clear();
    }
};

// One factorisation (+ optional inverse).  Launch sequence per block column k:
//     factor_step_kernel(k) = [look-ahead for k+1 | P1(k) + P2(k) | trtri row k-1]  ->  P3(k)
// and finally the trtri row n-1 alone.
struct FactorOpts {
    KSource src;            // src.K != nullptr: block columns >= 1 take their C tiles straight from K
    float* Y;               // nullptr: no triangular inverse
    TriReduce red;          // red.rpad != nullptr: fuse z-partials and Frobenius partials into trtri
};

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

// Timer classes: 0 = factor_step_kernel (P1 + P2 + trtri row k-1), 1 = factor_diag0_kernel (block column 0),
//                2 = potrf_trsm (P3),
//                3 = factor_step_kernel carrying only a trtri row (the last row; every row of volt_trtri_f32).
static void enqueue_step(const Group& g, int Np, int k, LaunchTimer* tm) {
    const int n = Np / TS, B = g.B;
    const TriReduce nored{nullptr, nullptr, nullptr, 0};
    const int itri = (g.o.Y && k > 0) ? k - 1 : -1;
    // k = 0 has no panel update: only the diagonal blocks of the prepared first column (t = 0 of n tiles
    // would waste n-1 idle workgroups per matrix, so the grid is cut to the diagonal tile alone)
    const int nupd = (k == 0) ? B : (n - k) * B;
    const int npre = (k >= 1 && k + 1 < n) ? B : 0;          // diagonal look-ahead workgroups (factor_step_kernel)
    if (tm) tm->begin(k == 0 ? 1 : 0);
    if (k == 0) {
        // decode_tile_batch(w, n - 0, B) would spread t over n tiles: launch with a private tile count of 1
        hipLaunchKernelGGL(factor_diag0_kernel, dim3(B), dim3(256), 0, g.s, g.A, g.Winv, g.info, Np);
    } else if (g.o.src.K) {
        hipLaunchKernelGGL(factor_step_kernel<true>, dim3(npre + nupd + (itri >= 0 ? (itri + 1) * B : 0)), dim3(256), 0, g.s,
                           g.A, g.Winv, g.o.Y, g.info, Np, k, itri, B, g.o.src, g.o.Y ? g.o.red : nored);
    } else {
        hipLaunchKernelGGL(factor_step_kernel<false>, dim3(npre + nupd + (itri >= 0 ? (itri + 1) * B : 0)), dim3(256), 0, g.s,
                           g.A, g.Winv, g.o.Y, g.info, Np, k, itri, B, g.o.src, g.o.Y ? g.o.red : nored);
    }
    if (tm) tm->end();
    if (k + 1 < n) {
        if (tm) tm->begin(2);
        hipLaunchKernelGGL(potrf_trsm_kernel, dim3((n - k - 1) * B), dim3(256), 0, g.s, g.A, g.Winv, Np, k, B);
        if (tm) tm->end();
    }
    if (k + 1 == n && g.o.Y) {
        if (tm) tm->begin(3);
        hipLaunchKernelGGL(factor_step_kernel<false>, dim3(n * B), dim3(256), 0, g.s, g.A, g.Winv, g.o.Y, g.info, Np,
                           -1, n - 1, B, g.o.src, g.o.red);
        if (tm) tm->end();
    }
}

static int run_factor(float* A, float* Winv, int* info, int B, int Np, hipStream_t s, const FactorOpts& o,
                      LaunchTimer* tm) {
    const int n = Np / TS;
    hipError_t e = hipMemsetAsync(info, 0, sizeof(int) * (size_t)B, s);
    if (e != hipSuccess) return (int)e;
    const Group g{A, Winv, info, o, B, s};
    for (int k = 0; k < n; ++k) enqueue_step(g, Np, k, tm);
    VOLT_LAUNCH_CHECK();
    return 0;
}

// ---- stage barriers vs. asynchrony ------------------------------------------------------------------
// With the whole batch in lockstep every launch ends in a tail: (n-k)*B tiles rarely fill a whole number of
// rounds of the 512 resident workgroups (measured: P1 at k = 23 runs 576 tiles = 1.125 rounds at 77 TF/s,
// k = 24 = 1.0 rounds at 131), and the next stage cannot start before the tail has drained.  A list-
// scheduling model of the measured tile times puts this at 32.4 ms per step against 26.8 ms for perfect
// packing.  Cutting the batch into G groups that run the SAME launch sequence on G streams, started a
// little apart, lets one group's tail overlap another group's next stage (model: 28.6 ms at G = 4).
// The streams are created once per device (library-lifetime, like a BLAS handle); a call forks from and
// joins back into the caller's stream with events, so the caller still sees ordinary stream semantics.
__global__ void delay_kernel(long long cycles) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
}

constexpr int MAX_GROUPS = 8;
struct StreamPool {
    hipStream_t aux[MAX_GROUPS - 1];
    bool ok = false;
};
static StreamPool* stream_pool() {
    static StreamPool pools[16];
    static std::once_flag once[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::call_once(once[dev], [dev]() {
        bool ok = true;
        for (int i = 0; i < MAX_GROUPS - 1; ++i)
            ok = ok && hipStreamCreateWithFlags(&pools[dev].aux[i], hipStreamNonBlocking) == hipSuccess;
        pools[dev].ok = ok;
    });
    return pools[dev].ok ? &pools[dev] : nullptr;
}

static int pick_groups(int B) {
    int want = 4;
    if (const char* e = getenv("VOLT_GROUPS")) want = atoi(e);
    if (want < 1) want = 1;
    if (want > MAX_GROUPS) want = MAX_GROUPS;
    while (want > 1 && (B % want != 0 || B / want < 8)) want >>= 1;   // keep whole-XCD groups of >= 8 matrices
    return want;
}

typedef void (*volt_group_post_fn)(void* ctx, int b0, int Bg, hipStream_t s);

static int run_factor_groups(float* A, float* Winv, int* info, int B, int Np, hipStream_t s, const FactorOpts& o,
                             volt_group_post_fn post = nullptr, void* post_ctx = nullptr) {
    const int n = Np / TS;
    const int G = pick_groups(B);
    StreamPool* pool = G > 1 ? stream_pool() : nullptr;
    if (G == 1 || !pool) {
        const int rc = run_factor(A, Winv, info, B, Np, s, o, nullptr);
        if (rc == 0 && post) post(post_ctx, 0, B, s);
        return rc;
    }
    hipError_t e = hipMemsetAsync(info, 0, sizeof(int) * (size_t)B, s);
    if (e != hipSuccess) return (int)e;
    const int Bg = B / G;
    const int64_t mat = (int64_t)Np * Np;
    Group grp[MAX_GROUPS];
    hipEvent_t fork, join[MAX_GROUPS];
    if ((e = hipEventCreateWithFlags(&fork, hipEventDisableTiming)) != hipSuccess) return (int)e;
    (void)hipEventRecord(fork, s);
    long long skew = 0;                                    // optional start skew between groups (measured: no effect)
    if (const char* ev = getenv("VOLT_GROUP_SKEW_US")) skew = (long long)(atof(ev) * 2400.0);
    for (int g = 0; g < G; ++g) {
        FactorOpts og = o;
        const int b0 = g * Bg;
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
        if (g > 0) {
            (void)hipStreamWaitEvent(grp[g].s, fork, 0);
            if (skew > 0) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, grp[g].s, skew * g);
        }
    }
    for (int k = 0; k < n; ++k)
        for (int g = 0; g < G; ++g) enqueue_step(grp[g], Np, k, nullptr);
    if (post)
        for (int g = 0; g < G; ++g) post(post_ctx, g * Bg, Bg, grp[g].s);
    for (int g = 1; g < G; ++g) {
        if ((e = hipEventCreateWithFlags(&join[g], hipEventDisableTiming)) != hipSuccess) {
            // out of events: fall back to a host-side join so that the caller's stream semantics still hold
            (void)hipStreamSynchronize(grp[g].s);
            continue;
        }
        (void)hipEventRecord(join[g], grp[g].s);
        (void)hipStreamWaitEvent(s, join[g], 0);
        (void)hipEventDestroy(join[g]);
    }
    (void)hipEventDestroy(fork);
    VOLT_LAUNCH_CHECK();
    return 0;
}

static int run_trtri(const float* A, const float* Winv, float* Y, int B, int Np, hipStream_t s, LaunchTimer* tm) {
    const int n = Np / TS;
    const TriReduce nored{nullptr, nullptr, nullptr, 0};
    for (int i = 0; i < n; ++i) {
        if (tm) tm->begin(3);
        hipLaunchKernelGGL(factor_step_kernel<false>, dim3((i + 1) * B), dim3(256), 0, s, const_cast<float*>(A),
                           const_cast<float*>(Winv), Y, nullptr, Np, -1, i, B, KSource{nullptr, 0, 0, nullptr, 0.f, 0},
                           nored);
        if (tm) tm->end();
    }
    VOLT_LAUNCH_CHECK();
    return 0;
}

// used by mll.hip
int volt_internal_factor(const float* K, int64_t ldk, int64_t bsk, const float* sigma2, float jitter, float* A,
                         float* Winv, float* Y, int* info, const float* rpad, float* zpart, float* frob, int B, int N,
                         void* stream, float* ms_host, int* launches_host, volt_group_post_fn post, void* post_ctx) {
    const int Np = volt_padded_n(N), n = Np / TS;
    hipStream_t s = (hipStream_t)stream;
    // block column 0 (no panel update there) is copied; the others are read from K inside P1
    hipLaunchKernelGGL(prepare_kernel, dim3(n, B), dim3(256), 0, s, K, ldk, bsk, sigma2, jitter, A, N, Np, 1);
    FactorOpts o{KSource{K, ldk, bsk, sigma2, jitter, N}, Y, TriReduce{rpad, zpart, frob, N}};
    if (!ms_host) return run_factor_groups(A, Winv, info, B, Np, s, o, post, post_ctx);
    LaunchTimer tm(s);
    const int rc = run_factor(A, Winv, info, B, Np, s, o, &tm);
    if (rc) return rc;
    tm.collect(ms_host, launches_host, 4);
    return 0;
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

int volt_tune_update_f32(float* A, int B, int Np, int k, int var, int reps, void* stream) {
    if (!A) return -1;
    if (B < 1) return -2;
    if (Np < TS || Np % TS) return -3;
    const int n = Np / TS;
    if (k < 1 || k >= n) return -4;
    if (var != 0) return -5;
    hipStream_t s = (hipStream_t)stream;
    const KSource none{nullptr, 0, 0, nullptr, 0.f, 0};
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(tune_update_kernel, dim3((n - k) * B), dim3(256), 0, s, A, Np, k, B, none);
    VOLT_LAUNCH_CHECK();
    return 0;
}

int volt_potrf_f32(float* A, float* Winv, int* info, int B, int Np, void* stream) {
    if (!A) return -1;
    if (!Winv) return -2;
    if (!info) return -3;
    if (B < 0) return -4;
    if (Np < TS || Np % TS) return -5;
    if (B == 0) return 0;
    FactorOpts o{KSource{nullptr, 0, 0, nullptr, 0.f, 0}, nullptr, TriReduce{nullptr, nullptr, nullptr, 0}};
    return run_factor_groups(A, Winv, info, B, Np, (hipStream_t)stream, o);
}

int volt_profile_factor_f32(const float* K, int64_t ldk, int64_t bsk, const float* sigma2, float* A, float* Winv,
                            float* Y, int* info, int B, int N, void* stream, float* ms_host, int* launches_host) {
    if (!K) return -1;
    if (ldk < N) return -2;
    if (!A) return -5;
    if (!Winv) return -6;
    if (!info) return -8;
    if (B < 1) return -9;
    if (N < 1) return -10;
    if (!ms_host) return -12;
    if (!launches_host) return -13;
    return volt_internal_factor(K, ldk, bsk, sigma2, 0.f, A, Winv, Y, info, nullptr, nullptr, nullptr, B, N, stream,
                                ms_host, launches_host, nullptr, nullptr);
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
