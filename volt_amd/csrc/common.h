// Shared device code for libvolt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VOLT_ABI_VERSION 2

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

template <int WL, int KK0 = 0, int KK1 = BK / 8>
__device__ __forceinline__ void mma_chunk(const float* __restrict__ buf, f32x16 (&acc)[4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const float* sA = buf;
    const float* sB = buf + TS * SLD;
#pragma unroll
    for (int kk = KK0; kk < KK1; ++kk) {
        const int ko = kk * 8 + 4 * lh;
        if (WL == 0) {
            const int wr = wave >> 1, wc = wave & 1;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(sA + (wr * 64 + l31) * SLD + ko);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(sA + (wr * 64 + 32 + l31) * SLD + ko);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sB + (wc * 64 + l31) * SLD + ko);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(sB + (wc * 64 + 32 + l31) * SLD + ko);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], b0[m], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], b1[m], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], b0[m], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], b1[m], acc[3], 0, 0, 0);
            }
        } else {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sB + (wave * 32 + l31) * SLD + ko);
            f32x4 a[4];
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
                a[tm] = *reinterpret_cast<const f32x4*>(sA + (tm * 32 + l31) * SLD + ko);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][m], b0[m], acc[tm], 0, 0, 0);
            }
        }
    }
}

// acc += A[0:128, 0:32*nchunks] * B[0:128, 0:32*nchunks]^T ; smem = GEMM_LDS_BYTES, 16-B aligned.
// Ends with a barrier: smem is free for reuse on return.
// The staging traffic rides in the shadow of the MFMAs: the LDS write of chunk c+1 and the global
// loads of chunk c+2 are issued half-way through the 64 MFMAs of chunk c (a 32x32x2 MFMA holds the
// pipe for 64 cycles; an in-order wave can slip a dozen other instructions behind each), so the
// only thing left at the per-chunk barrier is the barrier.
template <int WL>
__device__ __forceinline__ void gemm_nt_128(const float* __restrict__ A, int64_t lda,
                                               const float* __restrict__ B, int64_t ldb, int nchunks,
                                               f32x16 (&acc)[4], float* smem) {
    if (nchunks <= 0) return;
    // Two register stage sets: a chunk's loads are issued two chunks (128 MFMAs per wave) before its LDS
    // write, so a late HBM/L2 return no longer stalls the wave at the write (measured +x % at k = 16).
    StageRegs s0, s1;
    const StageAddr sa = stage_addr(A, lda, B, ldb);
    stage_load_buf(s0, sa, 0);
    stage_store(s0, smem);
    if (nchunks > 1) stage_load_buf(s0, sa, BK);
    if (nchunks > 2) stage_load_buf(s1, sa, 2 * BK);
    __syncthreads();
    int c = 0;
    for (; c + 1 < nchunks; c += 2) {
        float* b0 = smem;                      // chunk c (even) lives in buffer 0
        float* b1 = smem + STAGE_FLOATS;
        mma_chunk<WL, 0, BK / 16>(b0, acc);
        stage_store(s0, b1);                                            // chunk c+1
        if (c + 3 < nchunks) stage_load_buf(s0, sa, (c + 3) * BK);
        mma_chunk<WL, BK / 16, BK / 8>(b0, acc);
        __syncthreads();
        mma_chunk<WL, 0, BK / 16>(b1, acc);
        if (c + 2 < nchunks) stage_store(s1, b0);                       // chunk c+2
        if (c + 4 < nchunks) stage_load_buf(s1, sa, (c + 4) * BK);
        mma_chunk<WL, BK / 16, BK / 8>(b1, acc);
        __syncthreads();
    }
    if (c < nchunks) {                         // odd tail: chunk c sits in buffer 0
        mma_chunk<WL, 0, BK / 8>(smem, acc);
        __syncthreads();
    }
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

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
}

}  // namespace volt
