// Triangular solves with one right-hand side per matrix, ONE launch per solve.
//   reference call sites: torch.cholesky_solve at voltron/rollout_utils.py:36,44, VoltronGP.py:84-87,
//   VoltMagpie.py:88-91; gpytorch's inv_quad inside MultivariateNormal.log_prob (train_utils.py:249).
//
// L z = r (forward) or L^T x = z (backward) by 128-blocks, with the inverted diagonal blocks W_i the factorisation
// leaves behind:   z_i = W_i (r_i - sum_{m<i} L[i,m] z_m),      x_i = W_i^T (z_i - sum_{m>i} L[m,i]^T x_m).
// Block i of matrix b is one workgroup.  It streams its i (or n-1-i) tiles of L straight from HBM into registers
// -- every wave-instruction reads two 512-byte row segments, lanes along the columns -- and multiplies each by the
// solution block it depends on as soon as that block is published; the cross-lane reduction is deferred to the end
// of the block row (linearity), so a tile costs 16 loads and 64 FMAs per lane and nothing else.  The next tile's
// loads are issued before the wait for the current tile's dependency, so what remains on the critical path per
// hop is: flag seen -> acquire -> 512 B of solution -> FMAs -> LDS reduction -> W_i product -> publish.
//
// Inter-workgroup protocol: a solution block is 128 values -- the producer stores it WRITTEN THROUGH (agent-scope relaxed
// atomic stores = sc1), every storing thread drains, __syncthreads, lane 0 sets flag[b][i]; a consumer polls that one word
// relaxed from one lane, __syncthreads, and fetches the block with agent-scope (sc1) loads.  (Rounds 1-4 had plain stores
// behind an agent-scope release and plain loads behind an acquire -- Guideline 16 of cdna_hip_programming.md -- which puts an
// L2 write-back and an L2 / L1 invalidate on every hop of the chain: 8 x 4096 in fp32 4.1 us per hop against 3.0 now, in fp64
// 7.4 against 6.2; 64 x 2048 0.194 -> 0.139 ms per solve; scripts/bench_trsv.py.)  Block indices are handed out by an
// atomic ticket in dependency order (all matrices' block 0 first, ...), so a workgroup only ever waits for
// workgroups that started before it: no co-residency or dispatch-order assumption.  Every spin is bounded by wall
// clock; a time-out poisons the output with NaN instead of hanging and raises sync[1], the error word the caller can
// read back.  sync[] (ticket, error word, flags) is zeroed by the launcher.
#include "common.h"
#include "host.h"
#include "../../include/volt_hip.h"

namespace volt {

typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename T> struct V16;
template <> struct V16<float> { typedef f32x4 type; static constexpr int N = 4; };
template <> struct V16<double> { typedef f64x2 type; static constexpr int N = 2; };

__device__ __forceinline__ bool trsv_wait(const int* flag) {
    // one lane polls one word, relaxed, agent scope (bounded by wall clock, common.h); then one acquire for the workgroup
    bool ok = true;
    if (threadIdx.x == 0) ok = wait_nonzero(flag, 2);           // (no acquire: the block it announces is fetched with sc1 loads)
    return ok;      // meaningful in thread 0 only
}

__device__ __forceinline__ void trsv_publish(int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its stores
    __syncthreads();
    // (no release: the block went out written through -- sc1 -- and every storing thread has drained)
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Lane mapping of a 128 x CW sub-tile (CW = 32 VEC columns): lane = (half = l >> 5, lc = l & 31); wave w owns rows
// 32 w .. 32 w + 31; load s (0..15) of a lane is row 32 w + 2 s + half, columns VEC lc .. VEC lc + VEC - 1.
// PREW: the NH sub-tiles of W_i -- they depend on nothing -- are requested at the very start and kept in registers (64 / 128
// VGPRs), instead of behind the last dependency's tile: their load latency comes off every hop of the chain.  The host picks
// it for the forward fp64 solve of launches of at most two workgroups per CU (launch_trsv: where it was measured to pay).
template <typename T, bool TRANS, bool PREW>
__global__ __launch_bounds__(256) void trsv_kernel(const T* __restrict__ A, const T* __restrict__ Winv,
                                                   const T* rhs, T* out, int* __restrict__ sync, int Np, int B) {
    typedef typename V16<T>::type vec_t;
    constexpr int VEC = V16<T>::N, CW = 32 * VEC, NH = TS / CW;
    __shared__ T sP[TS * 33];             // forward: per-row partial sums [row][lc]; backward: [wave][column]
    __shared__ T sV[2][TS];               // the vector a sub-tile is multiplied with (double-buffered by parity)
    __shared__ int sTicket, sFail;
    const int n = Np / TS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, lc = lane & 31;
    if (tid == 0) {
        sTicket = atomicAdd(&sync[0], 1);
        sFail = 0;
    }
    __syncthreads();
    const int t = sTicket;
    const int i = TRANS ? n - 1 - t / B : t / B, b = t % B;
    int* flags = sync + 2 + b * n;
    const T* Ab = A + (int64_t)b * Np * Np;
    const T* Wi = Winv + ((int64_t)b * n + i) * TS * TS;
    const T* rb = rhs + (int64_t)b * Np;
    T* ob = out + (int64_t)b * Np;

    const int ndep = TRANS ? n - 1 - i : i;              // blocks this one depends on
    const int nl = ndep * NH, total = nl + NH;           // L sub-tiles, then the NH sub-tiles of W_i
    // sub-tile j < nl: dependency block m = (TRANS ? n-1 - j/NH : j/NH), column half h = j % NH
    auto sub_ptr = [&](int j) -> const T* {
        const int h = j % NH;
        if (j >= nl) return Wi + (int64_t)(32 * wave + half) * TS + h * CW + VEC * lc;
        const int m = TRANS ? n - 1 - j / NH : j / NH;
        return TRANS ? Ab + (int64_t)(m * TS + 32 * wave + half) * Np + (int64_t)i * TS + h * CW + VEC * lc
                     : Ab + (int64_t)(i * TS + 32 * wave + half) * Np + (int64_t)m * TS + h * CW + VEC * lc;
    };
    vec_t bufA[16], bufB[16];
    auto load = [&](vec_t (&dst)[16], int j) {
        const T* p = sub_ptr(j);
        const int64_t step = 2 * (j >= nl ? (int64_t)TS : (int64_t)Np);
#pragma unroll
        for (int s = 0; s < 16; ++s) dst[s] = *reinterpret_cast<const vec_t*>(p + s * step);
    };
    // forward: acc[s] = partial of row 32 w + 2 s + half over this lane's columns
    // backward: acc[h * VEC + e] = partial of column h CW + VEC lc + e over this lane's rows (16 slots: NH VEC <= 4 used)
    T acc[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = T(0);

    // reduce the partials of phase 1 into v = rhs_i - sum (128 values, in sV[0]); then clear acc
    auto finish_phase = [&](bool first) {
        if (!TRANS) {
#pragma unroll
            for (int s = 0; s < 16; ++s) sP[(32 * wave + 2 * s + half) * 33 + lc] = acc[s];
            __syncthreads();
            const int row = tid >> 1, part = tid & 1;
            T a = T(0);
#pragma unroll
            for (int q = 0; q < 16; ++q) a += sP[row * 33 + part * 16 + q];
            a += __shfl_xor(a, 1);
            if (part == 0) {
                if (first) sV[0][row] = rb[i * TS + row] - a;
                else sV[0][row] = a;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NH * VEC; ++q) acc[q] += __shfl_xor(acc[q], 32);
            if (half == 0) {
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) sP[wave * TS + h * CW + VEC * lc + e] = acc[h * VEC + e];
            }
            __syncthreads();
            if (tid < TS) {
                const T a = (sP[tid] + sP[TS + tid]) + (sP[2 * TS + tid] + sP[3 * TS + tid]);
                if (first) sV[0][tid] = rb[i * TS + tid] - a;
                else sV[0][tid] = a;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[s] = T(0);
    };

    auto consume = [&](const vec_t (&cur)[16], int j) {
        const int h = j % NH;
        if (j < nl) {
            const int dep = j / NH, m = TRANS ? n - 1 - dep : dep;
            if (h == 0) {                                   // first sub-tile of a new dependency block
                const bool ok = trsv_wait(flags + m);
                if (tid == 0 && !ok) {
                    sFail = 1;
                    atomicOr(&sync[1], 1);                  // the error word of the C ABI: this solve timed out
                }
                __syncthreads();                            // the acquire covers the workgroup
                if (tid < TS) sV[dep & 1][tid] = __hip_atomic_load(&ob[m * TS + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
            }
            const T* v = sV[dep & 1];
            if (!TRANS) {
                T vv[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) vv[e] = v[h * CW + VEC * lc + e];
#pragma unroll
                for (int s = 0; s < 16; ++s)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[s] += cur[s][e] * vv[e];
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const T xr = v[32 * wave + 2 * s + half];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[h * VEC + e] += cur[s][e] * xr;
                }
            }
        } else {
            if (j == nl) finish_phase(true);                // v = rhs_i - (sum over the dependency blocks)
            const T* v = sV[0];
            if (!TRANS) {                                   // z_i = W_i v
                T vv[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) vv[e] = v[h * CW + VEC * lc + e];
#pragma unroll
                for (int s = 0; s < 16; ++s)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[s] += cur[s][e] * vv[e];
            } else {                                        // x_i = W_i^T v
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const T xr = v[32 * wave + 2 * s + half];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[h * VEC + e] += cur[s][e] * xr;
                }
            }
        }
    };

    if constexpr (PREW) {
        vec_t bufW[NH][16];
#pragma unroll
        for (int h = 0; h < NH; ++h) load(bufW[h], nl + h);
        int j = 0;
        if (nl > 0) load(bufA, 0);
        for (; j + 1 < nl; j += 2) {
            load(bufB, j + 1);
            consume(bufA, j);
            if (j + 2 < nl) load(bufA, j + 2);
            consume(bufB, j + 1);
        }
        if (j < nl) consume(bufA, j);
#pragma unroll
        for (int h = 0; h < NH; ++h) consume(bufW[h], nl + h);
    } else {
        load(bufA, 0);
        int j = 0;
        for (; j + 1 < total; j += 2) {
            load(bufB, j + 1);
            consume(bufA, j);
            if (j + 2 < total) load(bufA, j + 2);
            consume(bufB, j + 1);
        }
        if (j < total) consume(bufA, j);
    }
    // sV[0] is still being read by slower waves of the W phase: the barriers inside finish_phase order that
    finish_phase(false);                                    // sV[0] = the solution block
    if (tid < TS) __hip_atomic_store(&ob[i * TS + tid], sFail ? (T)__builtin_nanf("") : sV[0][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    trsv_publish(flags + i);
}

template <typename T, bool TRANS>
static int launch_trsv(const T* A, const T* Winv, const T* rhs, T* out, T* scratch, int B, int Np, void* stream) {
    if (!A) return -1;
    if (!Winv) return -2;
    if (!rhs) return -3;
    if (!out) return -4;
    if (!scratch) return -5;
    if (B < 0 || B > 65535) return -6;
    if (Np < TS || Np % TS) return -7;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int n = Np / TS;
    // tickets + flags live in the caller's scratch ([B,Np] elements >= 2 + B n ints)
    int* sync = reinterpret_cast<int*>(scratch);
    hipError_t e = hipMemsetAsync(sync, 0, sizeof(int) * (size_t)(2 + (size_t)B * n), s);
    if (e != hipSuccess) return (int)e;
    // W_i ahead of the chain: measured (scripts/bench_trsv.py) 1 x 4096 fp64 forward 0.149 -> 0.114 ms, 8 x 4096 0.197 -> 0.146;
    // the fp64 transposed solve gets SLOWER (its column-wise partial sums and 128 more registers spill: 0.127 -> 0.165) and
    // the fp32 solves do not change (0.089 / 0.080): the forward fp64 solve only
    constexpr bool prew = sizeof(T) == 8 && !TRANS;
    if (prew && n * B <= 2 * tunables().cus)
        hipLaunchKernelGGL((trsv_kernel<T, TRANS, prew>), dim3(n * B), dim3(256), 0, s, A, Winv, rhs, out, sync, Np, B);
    else
        hipLaunchKernelGGL((trsv_kernel<T, TRANS, false>), dim3(n * B), dim3(256), 0, s, A, Winv, rhs, out, sync, Np, B);
    VOLT_LAUNCH_CHECK();
    return 0;
}

}  // namespace volt

using namespace volt;

extern "C" {

int volt_trsv_lower_f32(const float* A, const float* Winv, const float* rhs, float* out, float* scratch, int B,
                        int Np, void* stream) {
    return launch_trsv<float, false>(A, Winv, rhs, out, scratch, B, Np, stream);
}
int volt_trsv_lower_t_f32(const float* A, const float* Winv, const float* rhs, float* out, float* scratch, int B,
                          int Np, void* stream) {
    return launch_trsv<float, true>(A, Winv, rhs, out, scratch, B, Np, stream);
}
int volt_trsv_lower_f64(const double* A, const double* Winv, const double* rhs, double* out, double* scratch, int B,
                        int Np, void* stream) {
    return launch_trsv<double, false>(A, Winv, rhs, out, scratch, B, Np, stream);
}
int volt_trsv_lower_t_f64(const double* A, const double* Winv, const double* rhs, double* out, double* scratch, int B,
                          int Np, void* stream) {
    return launch_trsv<double, true>(A, Winv, rhs, out, scratch, B, Np, stream);
}

}  // extern "C"
