// Sequential posterior rollouts, bordered-Cholesky engine (7 hard part 3).
//   reference: voltron/rollout_utils.py:57-93 calling GeneratePrediction (:6-53) H times; every call
//   re-fills and re-factors S matrices of size (N+idx)^2 from scratch.
//
// Structure used (and only this): the train block of every sample's matrix is the same
// (train_stack_vol = log_vol_path.repeat(S,1), :72) and the cross block between the train points and
// the appended points is sample-independent.  So per series the host factors K_NN once (HIP potrf)
// and solves K_NN x = u once (volt_amd/rollout_engine.py: in closed form, or on the fp64
// factorisation); with rho = u'x (= q'q, q = L^-1 u) and tau = x'r_tr (= q'z_tr) every sample owns a small dense
// bordered problem of dimension idx <= H:
//     S_s   = C_s - rho 11'            (Schur complement of the appended points),  L_s = chol(S_s)
//     w_s   = L_s^-1 (k*_s - rho 1),   z_s = L_s^-1 (r_s - tau 1)
//     mean  = tau + w_s'z_s + m(x*),   var = k** - rho - w_s'w_s
// The new row of L_s at step idx+1 is w_s of step idx (the new column of K_tr is the previous k*),
// which is ordinary row-by-row Cholesky.  w_s itself is append-only for this kernel (see rollout_bordered_kernel);
// the full per-step re-substitution against the stored dense rows is kept as a cross-check mode.
//
// One wave per sample (a sample's recursion is sequential in idx and in the substitution index);
// 4 samples per workgroup, EWMA histories in LDS.  H <= 1024: a lane owns the entries 256 c + 4 lane + t of
// ceil(H / 256) chunks, so a stored row is read and written with 16-byte accesses, 1 KB per wave-instruction.
#include "common.h"
#include "host.h"
#include "../../include/volt_hip.h"

namespace volt {

// Sum of a double over the 64 lanes, result in every lane: DPP inside the 16-lane rows (the two halves of the double travel as
// two 32-bit DPP moves), then four readlane pairs -- the shape of common.h's wave_sum_f.  Round 6: the butterfly through
// __shfl_xor was six dependent ds_bpermute round trips per sum (2 x 6 LDS instructions, ~100 cycles each) on a path whose
// every step waits for its own mean: the step's dependent chain, not its instruction count, is what a rollout costs.
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ double dpp_add_d(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWS, 0xf, ROWS == 0xf);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWS, 0xf, ROWS == 0xf);
    return x + __hiloint2double(hi2, lo2);
}
// Inside the 16-lane rows as wave_sum_f does (quad_perm x 2, row_half_mirror, row_mirror), then ACROSS the rows with the two
// broadcast steps of the GFX9 wave reduction -- row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 -- which leave
// (r3 + r2) + (r1 + r0) in row 3: the value common.h's wave_sum_f forms as (r0 + r1) + (r2 + r3), bit for bit, in 6 DPP steps
// and one readlane (pair) instead of 4 + four readlanes and three scalar-operand adds.  The rollout kernel issues 212 VALU
// instructions per sample-step and is bound by exactly that count (profiles/r06/rollout_pmc.json): a step has four of these sums.
__device__ __forceinline__ double wave_sum_d(double x) {
    x = dpp_add_d<0xB1>(x);         // quad_perm [1,0,3,2]
    x = dpp_add_d<0x4E>(x);         // quad_perm [2,3,0,1]
    x = dpp_add_d<0x141>(x);        // row_half_mirror
    x = dpp_add_d<0x140>(x);        // row_mirror: every lane of a 16-lane row holds the row sum
    x = dpp_add_d<0x142, 0xa>(x);   // row_bcast:15 -> rows 1, 3
    x = dpp_add_d<0x143, 0xc>(x);   // row_bcast:31 -> rows 2, 3
    const int lo = __double2loint(x), hi = __double2hiint(x);
    return __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63));
}
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ float dpp_add_rows(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROWS, 0xf, ROWS == 0xf));
}
__device__ __forceinline__ float wave_sum_r(float x) {
    x = dpp_add_rows<0xB1>(x);
    x = dpp_add_rows<0x4E>(x);
    x = dpp_add_rows<0x141>(x);
    x = dpp_add_rows<0x140>(x);
    x = dpp_add_rows<0x142, 0xa>(x);
    x = dpp_add_rows<0x143, 0xc>(x);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}


// Mean of the appended point idx for the EWMA family (EWMA.py:20-37, :39-54, :74-91, :94-113, :116-135) from the
// per-sample histories in LDS (k train-tail values followed by the values appended so far).  ma1 = plain EMA at
// the new index, e2new = EMA(EMA) there (dewma / tewma).
// MODE >= 0: the mean mode at compile time (the kernel the product runs, one instance per mode: the tap loop carries no
// per-tap mode tests and only the sums its mode needs); MODE < 0: `mode` decides at run time (the other instances).
template <int MODE = -1>
__device__ __forceinline__ float family_mean(int mode_rt, int idx, int k, int lane, const float* sw, const float* hy,
                                             const float* he1, const float* he2, float ema_prev, float mr_theta,
                                             float mr_latent, float& ma1, float& e2new) {
    // (no contraction: the instances with the mode at compile time and at run time must round alike -- the re-substitution
    // cross-check is held to bitwise equality with the product's engine)
#pragma clang fp contract(off)
    const int mode = MODE >= 0 ? MODE : mode_rt;
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int j = lane; j < k; j += 64) {
        const double wj = (double)sw[j];
        a1 += wj * (double)hy[idx + j];
        if (mode == 1 || mode == 2) a2 += wj * (double)he1[idx + j];
        if (mode == 2) a3 += wj * (double)he2[idx + j];
    }
    ma1 = (float)wave_sum_d(a1);
    float mstar = ma1;
    e2new = 0.f;
    if (mode == 1 || mode == 2) {
        e2new = (float)wave_sum_d(a2);                       // EMA(EMA) at the new index
        if (mode == 1) mstar = 2.f * ma1 - e2new;
        else mstar = 3.f * ma1 - 3.f * e2new + (float)wave_sum_d(a3);
    } else if (mode == 3) {
        mstar = ma1 - mr_theta * (ema_prev - mr_latent);
    }
    return mstar;
}

struct RolloutParams {
    // per series g (G series), all device pointers
    const double* rho;       // [G]   q'q     (fp64: it is subtracted from the CumTrapz sums, see `base` below)
    const double* tau;       // [G]   q'z_tr
    const double* acc0;      // [G]   fp64 running CumTrapz sum through train point N-1 (full weights)
    const float* dx;         // [G]   x[1]-x[0]
    const float* hist_y;     // [G,k] padded train series tail  Y[N-k .. N-1]
    const float* hist_e1;    // [G,k] EMA tail    ema[N-k .. N-1]     (dewma / tewma)
    const float* hist_e2;    // [G,k] EMA(EMA) tail                    (tewma)
    const float* ema_prev;   // [G]   plain ema[N-1]                    (meanrevert)
    const float* mr_latent;  // [G]   latent mean of the mean module    (meanrevert)
    const float* latent;     // [G]   Rollouts' latent_mean (theta != NULL) or unused
    const float* w;          // [k]   EWMA taps
    const float* pred_vol;   // [G,S,H]
    const float* z;          // [G,S,H]
    float* samples;          // [G,S,H]
    float* Ls;               // [G,S,rollout_sample_floats(H)] scratch: packed strictly-lower rows of the per-sample factor
    int* info;               // [G,S] 0; +step (1-based) of the first non-positive pivot of L_s (a local jitter was
                             //       applied); -step if the predictive variance stayed <= 0 after the jitter ladder
    int G, S, H, k;
    int mean_mode;           // 0 ewma, 1 dewma, 2 tewma, 3 meanrevert, 4 given: hist_e1 = the mean at the H appended points [G,H]
    int use_theta;
    float theta, mr_theta, jitter;
};

// The strictly-lower rows of a sample's factor are stored PACKED: row a (a entries; its diagonal lives in registers)
// takes ceil(a/4) 16-byte slots right behind row a-1.  With one row per KiB (round 2's first layout) every row read
// fetched whole 128-byte lines of its own -- 1.18x the algorithmic bytes (PMC: 8.25e9 read requests, all 128-B) --
// and the store was twice as large.  row_off(a) = 4 * sum_{m<a} ceil(m/4) floats.
__device__ __forceinline__ int row_off(int a) {
    const int m = a - 1;                       // sum_{i=1}^{m} ceil(i/4) with m = 4 q + r:  (q+1)(2q + r)
    if (m <= 0) return 0;
    const int q = m >> 2, r = m & 3;
    return 4 * (q + 1) * (2 * q + r);
}
__host__ __device__ inline size_t rollout_sample_floats(int H) {
    const int m = H - 1, q = m >> 2, r = m & 3;
    return m <= 0 ? 4 : (size_t)4 * (q + 1) * (2 * q + r) + 4;    // rows 1 .. H-1 (+ one slot of slack)
}

// NC = 256-entry chunks per factor row (H <= 256 NC): a lane owns the entries 256 c + 4 lane + t, so a stored row is
// read and written with NC 16-byte accesses per lane, each wave-instruction covering 1 KB of contiguous memory.
//
// Two ways to get w_s at step idx:
//   RESUB = false  APPEND-ONLY (default).  For the volatility kernel the right-hand side prefix is step-invariant -- the
//      covariance between appended point a and ANY later point is U_s[a] (k(x*, x_a) = V[min(a, *)]) -- and the stored
//      rows never change, so w_s(idx) = [w_s(idx-1), new entry]: one dot product against the row appended last (it is
//      still in registers), nothing is re-read, no scratch.  O(idx) flops per step, O(H^2) per path.
//   RESUB = true   FULL RE-SUBSTITUTION against the stored rows at every step -- what a kernel without that invariance would
//      need.  O(idx^2) words streamed per step (H^3/6 * 4 B per path, HBM-bound).  Same arithmetic in the same order, so
//      the two produce BITWISE identical paths (tested); kept as the cross-check and as the measured cost of not
//      using the invariance (bench.py reports it as redundant bytes).
// MODE: the mean mode at compile time (0 .. 4), or -1 = p.mean_mode decides at run time.  The engine the product runs -- NC = 1,
// append-only -- has one instance per mode: round 5's mode 4 added a run-time test (and a load nobody else needs) to every step of
// every mode and cost 8 %.
template <int NC, bool RESUB, int MODE = -1>
__global__ __launch_bounds__(256) void rollout_bordered_kernel(RolloutParams p) {
    // Both instantiations must round identically (the test holds them to bitwise equality), so nothing here is left
    // to the compiler's choice of what to fuse: contraction off, every multiply-add that should be one is written as one.
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x * 4 + wave, g = blockIdx.y;
    const int H = p.H, k = p.k;
    const int mean_mode = MODE >= 0 ? MODE : p.mean_mode;
    const int hl = k + H;                                   // history length per level
    float* hy = lds + (size_t)wave * 3 * hl;
    float* he1 = hy + hl;
    float* he2 = he1 + hl;
    float* sw = lds + (size_t)4 * 3 * hl;                   // taps, shared by the 4 waves
    for (int j = threadIdx.x; j < k; j += 256) sw[j] = p.w[j];
    if (s < p.S) {
        for (int j = lane; j < k; j += 64) {
            hy[j] = p.hist_y[(size_t)g * k + j];
            he1[j] = (mean_mode == 1 || mean_mode == 2) ? p.hist_e1[(size_t)g * k + j] : 0.f;
            he2[j] = (mean_mode == 2) ? p.hist_e2[(size_t)g * k + j] : 0.f;
        }
    }
    __syncthreads();
    if (s >= p.S) return;

    const float tau = (float)p.tau[g], dx = p.dx[g], hdx = dx * 0.5f;
    const size_t row = ((size_t)g * p.S + s) * H;
    const float* pv = p.pred_vol + row;
    const float* zz = p.z + row;
    float* out = p.samples + row;
    float* Ls = RESUB ? p.Ls + ((size_t)g * p.S + s) * rollout_sample_floats(H) : nullptr;
    // base = U_s[N+a] - rho, carried in fp64: the entries of the Schur complement S_s = C_s - rho 11' are
    // ~ dx vol^2 (1e-4 .. 1e-9) on top of rho ~ V[N-1] ~ 1, so forming U_s and rho in fp32 first loses them
    // (62 of 80,000 paths at N = 4096 lost a pivot that way); the differences themselves are fine in fp32.
    double base = p.acc0[g] - p.rho[g];
    float ema_prev = (mean_mode == 3) ? p.ema_prev[g] : 0.f;
    const float mr_latent = (mean_mode == 3) ? p.mr_latent[g] : 0.f;
    int bad = 0;

    // RESUB: entry (c, t) of these arrays belongs to appended point b = 256 c + 4 lane + t
    float U[NC][4], rd[NC][4], zs[NC][4];                   // U_s - rho;  1 / L_s[b][b];  z_s[b]
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) U[c][t] = rd[c][t] = zs[c][t] = 0.f;
    // append-only: the running sums w_s'w_s and w_s'z_s, and the last appended point's (U_s - rho, 1 / L_ii, z_s)
    double ww_d = 0.0, wz_d = 0.0;
    float u_prev = 0.f, rell_prev = 0.f, z_prev = 0.f;

    // The step's inputs (pred_vol, z, a given mean) and its output travel in BLOCKS OF 64 STEPS, one per lane: one coalesced load /
    // store per 64 steps instead of two dependent global loads and a store inside every step -- each step then waited for its own
    // loads' round trips (and, behind the same counter, for the previous step's store): ~a third of a step's 2 us (round 6).  The
    // next block is requested while the current one is consumed; a step picks its values with v_readlane.
    const bool use_theta = p.use_theta != 0;
    const float lat_g = use_theta ? p.latent[g] : 0.f;
    const float* m4 = (mean_mode == 4) ? p.hist_e1 + (size_t)g * H : nullptr;
    float pv_blk = 0.f, zz_blk = 0.f, m4_blk = 0.f, out_blk = 0.f;
    float pv_nxt = (lane < H) ? pv[lane] : 0.f, zz_nxt = (lane < H) ? zz[lane] : 0.f;
    float m4_nxt = (m4 && lane < H) ? m4[lane] : 0.f;
    auto lane_pick = [](float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); };
    for (int idx = 0; idx < H; ++idx) {
        const int li = idx & 63;                             // (wave-uniform)
        if (li == 0) {
            pv_blk = pv_nxt;
            zz_blk = zz_nxt;
            m4_blk = m4_nxt;
            const int l2 = idx + 64 + lane;
            pv_nxt = (l2 < H) ? pv[l2] : 0.f;
            zz_nxt = (l2 < H) ? zz[l2] : 0.f;
            m4_nxt = (m4 && l2 < H) ? m4[l2] : 0.f;
        }
        float wv[NC][4];
        if constexpr (RESUB) {
            // ---- w_s = L_s^-1 (U_s - rho): row-oriented forward substitution against ALL stored rows ----
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t) wv[c][t] = U[c][t];
            // Rows are streamed in groups of 4 with the next group's loads issued before the current group is
            // consumed: the substitution is a dependent chain (row a needs w[a-1]) but the addresses are not, so 8
            // rows per wave stay in flight and the stream runs at memory bandwidth, not one latency per row.
            float cur[4][NC][4], nxt[4][NC][4];
            auto load_rows = [&](float (&dst)[4][NC][4], int a0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a = a0 + r;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const int b0 = 256 * c + 4 * lane;
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if (a < idx && b0 < a) v = *reinterpret_cast<const f32x4*>(Ls + row_off(a) + b0);
#pragma unroll
                        for (int t = 0; t < 4; ++t) dst[r][c][t] = (b0 + t < a) ? v[t] : 0.f;   // beyond b < a: never written
                    }
                }
            };
            load_rows(cur, 1);                               // row 0 has no off-diagonal part
            for (int a0 = 1; a0 < idx; a0 += 4) {
                load_rows(nxt, a0 + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a = a0 + r;
                    if (a < idx) {                           // wave-uniform
                        float part = 0.f;
#pragma unroll
                        for (int c = 0; c < NC; ++c)
#pragma unroll
                            for (int t = 0; t < 4; ++t)                                            // zero beyond b < a
                                part = __fmaf_rn(cur[r][c][t], __fmul_rn(wv[c][t], rd[c][t]), part);
                        const float dot = wave_sum_r(part);
#pragma unroll
                        for (int c = 0; c < NC; ++c)
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                if (256 * c + 4 * lane + t == a) wv[c][t] -= dot;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < NC; ++c)
#pragma unroll
                        for (int t = 0; t < 4; ++t) cur[r][c][t] = nxt[r][c][t];
            }
        }
        float ww = 0.f, wz = 0.f;
        if constexpr (RESUB) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int b = 256 * c + 4 * lane + t;
                    wv[c][t] = (b < idx) ? __fmul_rn(wv[c][t], rd[c][t]) : 0.f;    // now wv = w_s
                    ww = __fmaf_rn(wv[c][t], wv[c][t], ww);
                    wz = __fmaf_rn(wv[c][t], zs[c][t], wz);
                }
            ww = wave_sum_r(ww);
            wz = wave_sum_r(wz);
        } else {
            // ---- append-only: entries b < idx-1 of w_s stand; the new one, b = a = idx-1, is what the full substitution would
            // take against row a: (U_a - rho - sum_{b<a} L_s[a][b] w_b) / L_s[a][a] with L_s[a][b] = w_b -- the sum is w_s'w_s of
            // the step before.  So the recursion needs NO vector state: per step one new entry from three numbers of the previous
            // step, and w_s'w_s / w_s'z_s as running sums (fp64: exact products, one rounding per step).  Round 6: rounds 2 - 5
            // kept w_s, 1 / L_ii and z_s spread over the lanes and took three wave sums and ~40 masked element updates per step
            // to get the same numbers -- 110 of the step's 212 VALU instructions.
            if (idx >= 1) {
                const float dot = (float)ww_d;
                const float w_a = __fmul_rn(u_prev - dot, rell_prev);
                ww_d = __fma_rn((double)w_a, (double)w_a, ww_d);
                wz_d = __fma_rn((double)w_a, (double)z_prev, wz_d);
            }
            ww = (float)ww_d;
            wz = (float)wz_d;
        }

        // ---- mean of the new point: EWMA family on the stacked series (EWMA.py:20-37) ----------
        // (mode 4: a mean that is a function of x alone -- constant / linear / log-linear, the weather driver's default,
        // experiments/weather/GPGenerator.py:68-82 -- is history-free: the host evaluated it at the test points)
        float ma1 = 0.f, e2new = 0.f;
        const float mstar = (mean_mode == 4) ? lane_pick(m4_blk, li)
                                             : family_mean<MODE>(mean_mode, idx, k, lane, sw, hy, he1, he2, ema_prev, p.mr_theta,
                                                                 mr_latent, ma1, e2new);

        // ---- conditional and draw (rollout_utils.py:36-53) --------------------------------------
        const float v = lane_pick(pv_blk, li);
        const float v2 = v * v;
        const float kss = (float)(base + (double)__fmul_rn(hdx, v2));      // k** - rho; last CumTrapz weight halved
        float pm = tau + wz + mstar;
        if (use_theta) pm = __fmaf_rn(-p.theta, pm - lat_g, pm);
        float pvar = kss - ww;
        if (!(pvar > 0.f)) {                                 // psd_safe_cholesky(pred_cov, jitter) ladder
            float jit = p.jitter;
            int tries = 0;
            while (!(pvar + jit > 0.f) && tries < 2) { jit *= 10.f; ++tries; }
            if (pvar + jit > 0.f) pvar += jit;
            else { if (bad >= 0) bad = -(idx + 1); pvar = 0.f; }     // ladder exhausted: the reference raises NotPSDError
        }
        const float smp = __fmaf_rn(sqrtf(pvar), lane_pick(zz_blk, li), pm);
        out_blk = (lane == li) ? smp : out_blk;
        if (li == 63 || idx == H - 1) {                      // the block's samples out, one per lane
            const int l = (idx & ~63) + lane;
            if (l <= idx) out[l] = out_blk;
        }

        // ---- append the point to the conditioning set --------------------------------------------
        base += (double)__fmul_rn(dx, v2);                   // full weight from now on
        const float Unew = (float)base;
        float d2 = Unew - ww;                                // next pivot of L_s
        if (!(d2 > 0.f)) {
            if (!bad) bad = idx + 1;
            d2 = fmaxf(p.jitter, 1e-12f);
        }
        // 1 / L_s[idx][idx] straight from the hardware reciprocal square root (1 ulp), the pivot from it: the IEEE square root
        // followed by an IEEE division was 26 of the step's 212 VALU instructions
        const float rell = __builtin_amdgcn_rsqf(d2), ell = __fmul_rn(d2, rell);
        const float znew = ((smp - mstar) - tau - wz) * rell;
        // row idx of L_s = [w_s, ell]: the off-diagonal part goes to the packed store (re-substitution mode only),
        // ell stays in registers (rd)
        if constexpr (RESUB) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int b0 = 256 * c + 4 * lane;
                f32x4 rowv;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int b = b0 + t;
                    rowv[t] = (b < idx) ? wv[c][t] : ((b == idx) ? ell : 0.f);
                    if (b == idx) {
                        U[c][t] = Unew;
                        rd[c][t] = rell;
                        zs[c][t] = znew;
                    }
                }
                if (b0 < idx) *reinterpret_cast<f32x4*>(Ls + row_off(idx) + b0) = rowv;
            }
        } else {
            u_prev = Unew;                                   // (U_s - rho) of the appended point, its 1 / L_ii and its z_s: all the next step needs
            rell_prev = rell;
            z_prev = znew;
        }
        if (lane == 0) {                                     // (the histories a mode never reads are not kept)
            hy[k + idx] = smp;
            if (mean_mode == 1 || mean_mode == 2) he1[k + idx] = ma1;
            if (mean_mode == 2) he2[k + idx] = e2new;
        }
        ema_prev = ma1;
        if constexpr (RESUB) {
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();                           // this wave's own global writes (the factor's rows) before its later reads
        } else {
            // LDS only, and only this wave's: its LDS operations execute in order -- keep the compiler from moving accesses
            // across, no wait is needed (a __threadfence_block here also waited for the sample's store every step)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    if (lane == 0) p.info[(size_t)g * p.S + s] = bad;
}


// ---------------------------------------------------------------------------------------------------------
// ONE LANE PER PATH (round 6).  With the append-only recursion reduced to scalars (above: per step one new entry of w_s from three
// numbers of the step before, w_s'w_s and w_s'z_s as running sums) nothing in a step but the moving-average mean is a vector
// operation -- and a wave per path spent 63 of its 64 lanes repeating uniform arithmetic: ~120 wave-instructions per PATH-step.
// Here a wave carries 64 paths of one series, every lane its own recursion; the mean's window lives in LDS as a ring
// [k][64 lanes] per level (conflict-free: a tap is one ds_read for all 64 paths), the taps as doubles beside it, and the
// step's inputs / outputs move as 16-byte groups of four steps per lane, requested a group ahead.  ~150 wave-instructions per
// 64 path-steps.  The arithmetic per path is the wave-per-path engine's, operation for operation, except the order in which
// the k tap products are added (serially here, by lanes and a DPP tree there; both in fp64): the two agree to the last bit
// or the last bit but one of the mean, and are held to 1e-6 (tests/test_gpu_edge.py).
// LDS: levels * k * 256 B + 8 k; shapes that do not fit 150 KB (k = 400 with tewma's three levels) keep the wave-per-path engine.
template <int MODE, bool VEC>
__global__ __launch_bounds__(64) void rollout_lane_kernel(RolloutParams p) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LEVELS = MODE == 2 ? 3 : (MODE == 1 ? 2 : (MODE == 4 ? 0 : 1));
    const int lane = threadIdx.x;
    const int g = blockIdx.y, H = p.H, k = p.k;
    const int s_raw = blockIdx.x * 64 + lane;
    const bool active = s_raw < p.S;
    const int s = active ? s_raw : p.S - 1;                  // (idle lanes shadow the last path: every load is valid, nothing is stored)
    double* swd = reinterpret_cast<double*>(lds);            // [k] taps
    float* r0 = lds + 2 * k;                                 // [k][64] the stacked series
    float* r1 = r0 + (LEVELS > 1 ? 64 * k : 0);             // [k][64] its EMA           (dewma / tewma)
    float* r2 = r1 + (LEVELS > 2 ? 64 * k : 0);             // [k][64] EMA of the EMA    (tewma)
    if (LEVELS > 0) {
        for (int j = lane; j < k; j += 64) swd[j] = (double)p.w[j];
        for (int j = 0; j < k; ++j) {
            r0[64 * j + lane] = p.hist_y[(size_t)g * k + j];
            if (LEVELS > 1) r1[64 * j + lane] = p.hist_e1[(size_t)g * k + j];
            if (LEVELS > 2) r2[64 * j + lane] = p.hist_e2[(size_t)g * k + j];
        }
    }
    __syncthreads();
    const float tau = (float)p.tau[g], dx = p.dx[g], hdx = dx * 0.5f;
    const size_t row = ((size_t)g * p.S + s) * H;
    const float* pv = p.pred_vol + row;
    const float* zz = p.z + row;
    float* out = p.samples + row;
    double base = p.acc0[g] - p.rho[g];                      // U_s - rho, fp64 (see rollout_bordered_kernel)
    float ema_prev = (MODE == 3) ? p.ema_prev[g] : 0.f;
    const float mr_latent = (MODE == 3) ? p.mr_latent[g] : 0.f;
    const bool use_theta = p.use_theta != 0;
    const float lat_g = use_theta ? p.latent[g] : 0.f;
    const float* m4 = (MODE == 4) ? p.hist_e1 + (size_t)g * H : nullptr;
    int bad = 0;
    double ww_d = 0.0, wz_d = 0.0;
    float u_prev = 0.f, rell_prev = 0.f, z_prev = 0.f;
    f32x4 pv4 = {0.f, 0.f, 0.f, 0.f}, zz4 = pv4, o4 = pv4, pvn = pv4, zzn = pv4;
    float pv1 = 0.f, zz1 = 0.f;
    if (VEC) {
        pvn = *reinterpret_cast<const f32x4*>(pv);
        zzn = *reinterpret_cast<const f32x4*>(zz);
    } else {
        pv1 = pv[0];
        zz1 = zz[0];
    }
    int pos = 0;                                             // idx mod k: the ring slot of the window's oldest value
    for (int idx = 0; idx < H; ++idx) {
        const int q = idx & 3;
        float v, zi;
        if (VEC) {
            if (q == 0) {
                pv4 = pvn;
                zz4 = zzn;
                if (idx + 4 < H) {
                    pvn = *reinterpret_cast<const f32x4*>(pv + idx + 4);
                    zzn = *reinterpret_cast<const f32x4*>(zz + idx + 4);
                }
            }
            v = q == 0 ? pv4[0] : (q == 1 ? pv4[1] : (q == 2 ? pv4[2] : pv4[3]));
            zi = q == 0 ? zz4[0] : (q == 1 ? zz4[1] : (q == 2 ? zz4[2] : zz4[3]));
        } else {
            v = pv1;
            zi = zz1;
            if (idx + 1 < H) {
                pv1 = pv[idx + 1];
                zz1 = zz[idx + 1];
            }
        }
        // ---- the new entry of w_s and the running sums (rollout_bordered_kernel, append-only)
        if (idx >= 1) {
            const float dot = (float)ww_d;
            const float w_a = __fmul_rn(u_prev - dot, rell_prev);
            ww_d = __fma_rn((double)w_a, (double)w_a, ww_d);
            wz_d = __fma_rn((double)w_a, (double)z_prev, wz_d);
        }
        const float ww = (float)ww_d, wz = (float)wz_d;
        // ---- mean of the new point: the window [idx, idx + k) of the stacked series against the taps (EWMA.py:20-37), tap j
        // on ring slot (pos + j) mod k
        float ma1 = 0.f, e2new = 0.f, mstar;
        if (MODE == 4) {
            mstar = m4[idx];
        } else {
            double a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int j = 0;
            for (int m = pos; m < k; ++m, ++j) {
                const double wj = swd[j];
                a1 = __fma_rn(wj, (double)r0[64 * m + lane], a1);
                if (LEVELS > 1) a2 = __fma_rn(wj, (double)r1[64 * m + lane], a2);
                if (LEVELS > 2) a3 = __fma_rn(wj, (double)r2[64 * m + lane], a3);
            }
            for (int m = 0; m < pos; ++m, ++j) {
                const double wj = swd[j];
                a1 = __fma_rn(wj, (double)r0[64 * m + lane], a1);
                if (LEVELS > 1) a2 = __fma_rn(wj, (double)r1[64 * m + lane], a2);
                if (LEVELS > 2) a3 = __fma_rn(wj, (double)r2[64 * m + lane], a3);
            }
            ma1 = (float)a1;
            mstar = ma1;
            if (MODE == 1 || MODE == 2) {
                e2new = (float)a2;
                if (MODE == 1) mstar = 2.f * ma1 - e2new;
                else mstar = 3.f * ma1 - 3.f * e2new + (float)a3;
            } else if (MODE == 3) {
                mstar = ma1 - p.mr_theta * (ema_prev - mr_latent);
            }
        }
        // ---- conditional and draw (rollout_utils.py:36-53)
        const float v2 = v * v;
        const float kss = (float)(base + (double)__fmul_rn(hdx, v2));
        float pm = tau + wz + mstar;
        if (use_theta) pm = __fmaf_rn(-p.theta, pm - lat_g, pm);
        float pvar = kss - ww;
        if (!(pvar > 0.f)) {                                 // psd_safe_cholesky(pred_cov, jitter) ladder
            float jit = p.jitter;
            int tries = 0;
            while (!(pvar + jit > 0.f) && tries < 2) { jit *= 10.f; ++tries; }
            if (pvar + jit > 0.f) pvar += jit;
            else { if (bad >= 0) bad = -(idx + 1); pvar = 0.f; }
        }
        const float smp = __fmaf_rn(sqrtf(pvar), zi, pm);
        if (VEC) {
            if (q == 0) o4[0] = smp; else if (q == 1) o4[1] = smp; else if (q == 2) o4[2] = smp; else o4[3] = smp;
            if (q == 3 && active) *reinterpret_cast<f32x4*>(out + idx - 3) = o4;
        } else if (active) {
            out[idx] = smp;
        }
        // ---- append the point
        base += (double)__fmul_rn(dx, v2);
        const float Unew = (float)base;
        float d2 = Unew - ww;
        if (!(d2 > 0.f)) {
            if (!bad) bad = idx + 1;
            d2 = fmaxf(p.jitter, 1e-12f);
        }
        const float rell = __builtin_amdgcn_rsqf(d2);
        u_prev = Unew;
        rell_prev = rell;
        z_prev = ((smp - mstar) - tau - wz) * rell;
        if (LEVELS > 0) {                                    // the new values take the slot of the window's oldest
            r0[64 * pos + lane] = smp;
            if (LEVELS > 1) r1[64 * pos + lane] = ma1;
            if (LEVELS > 2) r2[64 * pos + lane] = e2new;
            pos = pos + 1 == k ? 0 : pos + 1;
        }
        ema_prev = ma1;
    }
    if (active) p.info[(size_t)g * p.S + s] = bad;
}


// ---------------------------------------------------------------------------------------------------------
// Rollouts of a GP whose kernel does not depend on the sample (voltron/rollout_utils.py:95-115, nonvol_rollouts):
// every sample shares one factorisation of the (N+H)^2 matrix, and the new entry of L^-1 r for an appended draw
// is gamma_j z_j whatever the history, so the GP part of every path is e = c + M z (one small GEMM on the host
// side, volt_amd/rollout_engine.py).  What stays sequential per sample is the moving-average mean of the
// stacked series: samples[idx] = mean_s(idx) + e[idx], with mean_s over the sample's own earlier draws.
struct SharedParams {
    const float *hist_y, *hist_e1, *hist_e2, *ema_prev, *mr_latent, *w, *e;
    float* samples;
    int G, S, H, k, mean_mode;
    float mr_theta;
};

__global__ __launch_bounds__(256) void rollout_shared_kernel(SharedParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x * 4 + wave, g = blockIdx.y;
    const int H = p.H, k = p.k, hl = k + H;
    float* hy = lds + (size_t)wave * 3 * hl;
    float* he1 = hy + hl;
    float* he2 = he1 + hl;
    float* sw = lds + (size_t)4 * 3 * hl;
    for (int j = threadIdx.x; j < k; j += 256) sw[j] = p.w[j];
    if (s < p.S) {
        for (int j = lane; j < k; j += 64) {
            hy[j] = p.hist_y[(size_t)g * k + j];
            he1[j] = (p.mean_mode == 1 || p.mean_mode == 2) ? p.hist_e1[(size_t)g * k + j] : 0.f;
            he2[j] = (p.mean_mode == 2) ? p.hist_e2[(size_t)g * k + j] : 0.f;
        }
    }
    __syncthreads();
    if (s >= p.S) return;
    const size_t row = ((size_t)g * p.S + s) * H;
    float ema_prev = (p.mean_mode == 3) ? p.ema_prev[g] : 0.f;
    for (int idx = 0; idx < H; ++idx) {
        float ma1, e2new;
        const float mstar = family_mean(p.mean_mode, idx, k, lane, sw, hy, he1, he2, ema_prev, p.mr_theta,
                                        (p.mean_mode == 3) ? p.mr_latent[g] : 0.f, ma1, e2new);
        const float smp = mstar + p.e[row + idx];
        if (lane == 0) {
            p.samples[row + idx] = smp;
            hy[k + idx] = smp;
            he1[k + idx] = ma1;
            he2[k + idx] = e2new;
        }
        ema_prev = ma1;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
}

}  // namespace volt

extern "C" {

size_t volt_rollout_scratch_bytes(int G, int S, int H) {
    if (G <= 0 || S <= 0 || H <= 0) return 0;
    return (size_t)G * S * volt::rollout_sample_floats(H) * sizeof(float);      // packed strictly-lower rows, 16-byte slots
}

int volt_rollout_bordered_f32(const double* rho, const double* tau, const double* acc0, const float* dx,
                              const float* hist_y, const float* hist_e1, const float* hist_e2, const float* ema_prev,
                              const float* mr_latent, const float* latent, const float* w, const float* pred_vol,
                              const float* z, float* samples, float* scratch, int* info, int G, int S, int H, int k,
                              int mean_mode, int use_theta, float theta, float mr_theta, float jitter, void* stream) {
    using namespace volt;
    if (!rho) return -1;
    if (!tau) return -2;
    if (!acc0) return -3;
    if (!dx) return -4;
    if (!hist_y) return -5;
    if (!w) return -11;
    if (!pred_vol) return -12;
    if (!z) return -13;
    if (!samples) return -14;
    if (!info) return -16;
    if (G < 0) return -17;
    if (S < 0) return -18;
    if (H < 1 || H > VOLT_ROLLOUT_MAX_H) return -19;
    if (k < 1 || k > 2048) return -20;
    if (mean_mode < 0 || mean_mode > 4) return -21;
    if ((mean_mode == 1 || mean_mode == 2 || mean_mode == 4) && !hist_e1) return -6;
    if (mean_mode == 2 && !hist_e2) return -7;
    if (mean_mode == 3 && (!ema_prev || !mr_latent)) return -8;
    if (use_theta && !latent) return -10;
    if (G == 0 || S == 0) return 0;
    RolloutParams p{rho, tau, acc0, dx, hist_y, hist_e1, hist_e2, ema_prev, mr_latent, latent, w, pred_vol, z,
                    samples, scratch, info, G, S, H, k, mean_mode, use_theta, theta, mr_theta, jitter};
    const size_t lds = ((size_t)4 * 3 * (k + H) + k) * sizeof(float);
    if (lds > 160 * 1024) return -20;
    const dim3 grid((S + 3) / 4, G);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
#define VOLT_ROLLOUT_LAUNCH1(NC, RS)                                                                                   \
    do {                                                                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(rollout_bordered_kernel<NC, RS>),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
        if (e != hipSuccess) return (int)e;                                                                            \
        hipLaunchKernelGGL((rollout_bordered_kernel<NC, RS>), grid, dim3(256), lds, s, p);                           \
    } while (0)
#define VOLT_ROLLOUT_LAUNCH(NC)                                                                                        \
    do {                                                                                                               \
        if (scratch) VOLT_ROLLOUT_LAUNCH1(NC, true);                                                                   \
        else VOLT_ROLLOUT_LAUNCH1(NC, false);                                                                          \
    } while (0)
    // the engine the product runs: one LANE per path, whenever the mean's window fits the LDS ring (levels * k * 256 B)
    const int levels = mean_mode == 2 ? 3 : (mean_mode == 1 ? 2 : (mean_mode == 4 ? 0 : 1));
    const size_t lane_lds = (size_t)levels * k * 256 + (size_t)k * 8;
    if (!scratch && lane_lds <= 150 * 1024 && tunables().rollout_lane != 0) {
        const bool vec = (H & 3) == 0 && (((uintptr_t)pred_vol | (uintptr_t)z | (uintptr_t)samples) & 15) == 0;
        const dim3 lgrid((S + 63) / 64, G);
#define VOLT_ROLLOUT_LANE1(M, V)                                                                                       \
    do {                                                                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(rollout_lane_kernel<M, V>),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lane_lds);                          \
        if (e != hipSuccess) return (int)e;                                                                            \
        hipLaunchKernelGGL((rollout_lane_kernel<M, V>), lgrid, dim3(64), lane_lds, s, p);                            \
    } while (0)
#define VOLT_ROLLOUT_LANE(M) do { if (vec) VOLT_ROLLOUT_LANE1(M, true); else VOLT_ROLLOUT_LANE1(M, false); } while (0)
        switch (mean_mode) {
            case 0: VOLT_ROLLOUT_LANE(0); break;
            case 1: VOLT_ROLLOUT_LANE(1); break;
            case 2: VOLT_ROLLOUT_LANE(2); break;
            case 3: VOLT_ROLLOUT_LANE(3); break;
            default: VOLT_ROLLOUT_LANE(4); break;
        }
#undef VOLT_ROLLOUT_LANE
#undef VOLT_ROLLOUT_LANE1
        VOLT_LAUNCH_CHECK();
        return 0;
    }
    if (H <= 256 && !scratch) {                              // a wave per path (windows too long for the ring): one instance per mean mode
#define VOLT_ROLLOUT_LAUNCH_MODE(M)                                                                                    \
    do {                                                                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(rollout_bordered_kernel<1, false, M>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
        if (e != hipSuccess) return (int)e;                                                                            \
        hipLaunchKernelGGL((rollout_bordered_kernel<1, false, M>), grid, dim3(256), lds, s, p);                      \
    } while (0)
        switch (mean_mode) {
            case 0: VOLT_ROLLOUT_LAUNCH_MODE(0); break;
            case 1: VOLT_ROLLOUT_LAUNCH_MODE(1); break;
            case 2: VOLT_ROLLOUT_LAUNCH_MODE(2); break;
            case 3: VOLT_ROLLOUT_LAUNCH_MODE(3); break;
            default: VOLT_ROLLOUT_LAUNCH_MODE(4); break;
        }
#undef VOLT_ROLLOUT_LAUNCH_MODE
    } else if (H <= 256) VOLT_ROLLOUT_LAUNCH(1);
    else if (H <= 512) VOLT_ROLLOUT_LAUNCH(2);
    else VOLT_ROLLOUT_LAUNCH(4);
#undef VOLT_ROLLOUT_LAUNCH
#undef VOLT_ROLLOUT_LAUNCH1
    VOLT_LAUNCH_CHECK();
    return 0;
}

int volt_rollout_shared_f32(const float* hist_y, const float* hist_e1, const float* hist_e2, const float* ema_prev,
                            const float* mr_latent, const float* w, const float* e, float* samples, int G, int S, int H,
                            int k, int mean_mode, float mr_theta, void* stream) {
    using namespace volt;
    if (!hist_y) return -1;
    if (!w) return -6;
    if (!e) return -7;
    if (!samples) return -8;
    if (G < 0) return -9;
    if (S < 0) return -10;
    if (H < 1 || H > 4096) return -11;
    if (k < 1 || k > 2048) return -12;
    if (mean_mode < 0 || mean_mode > 3) return -13;
    if ((mean_mode == 1 || mean_mode == 2) && !hist_e1) return -2;
    if (mean_mode == 2 && !hist_e2) return -3;
    if (mean_mode == 3 && (!ema_prev || !mr_latent)) return -4;
    if (G == 0 || S == 0) return 0;
    SharedParams p{hist_y, hist_e1, hist_e2, ema_prev, mr_latent, w, e, samples, G, S, H, k, mean_mode, mr_theta};
    const size_t lds = ((size_t)4 * 3 * (k + H) + k) * sizeof(float);
    if (lds > 160 * 1024) return -11;
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(rollout_shared_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return (int)err;
    hipLaunchKernelGGL(rollout_shared_kernel, dim3((S + 3) / 4, G), dim3(256), lds, (hipStream_t)stream, p);
    VOLT_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
