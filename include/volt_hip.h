/* volt_hip.h -- C ABI of libvolt_hip.so: Volt's exact-GP hot path on MI355X (gfx950).
 *
 * This is the operator-level boundary of SURVEY.md 8(b).  The reference has no native layer:
 * its hot path is a sequence of ATen calls issued from Python (file:line under the g-benton/Volt
 * tree).  Each entry point below names the reference call site(s) it replaces.  A maintainer binds
 * them with ctypes (INTEGRATION.md shows the stub); volt_amd/_lib.py is that binding.
 *
 * Conventions (all entry points):
 *   - Plain pointers and sizes only.  Every pointer except `stream` is a DEVICE pointer owned by the
 *     caller (e.g. torch `tensor.data_ptr()`).  The library never allocates, frees or retains
 *     caller-visible memory; scratch comes from the caller via the *_workspace_bytes queries.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); work is
 *     enqueued asynchronously and the call returns immediately.  Re-entrant across streams.  Library state:
 *     one pool of auxiliary streams and fork/join events per device, created on first use and kept for the
 *     life of the process (the batched factorisation runs groups of matrices on them, forked from and joined
 *     back into `stream`); calls on one device serialise on the host while they ENQUEUE, never on the device.
 *     And HOST-side only: a cache of launch schedules per (batch size, block columns, inverse?) -- a few hundred KB of
 *     tables in pinned host memory (3 <= B <= 64, csrc/sched.h), a pure function of the shape.  The library owns NO
 *     device memory and keeps NO record of caller workspaces: volt_mll_workspace_init_f32 / volt_potrf_workspace_init_f32
 *     copy a table into the caller's scratch, and the caller states that it did so with the VOLT_WS_INITIALISED flag of
 *     the entry points that take a workspace.  Without the flag a call runs the table-free schedules; with it every
 *     table-driven launch checks the header in the scratch first, so a region that was never initialised, or was
 *     overwritten since, is reported (info = INT_MIN + 1), never followed.
 *   - Return value: 0 = enqueued; -k = argument k (1-based) is invalid; >0 = hipError_t of a
 *     failed launch.  A matrix that is not positive definite is NOT an error return: LAPACK-style
 *     `info[b]` (0, or the 1-based index of the first non-positive / NaN pivot) is written to a
 *     caller buffer and the jitter-retry policy (gpytorch psd_safe_cholesky, used at
 *     voltron/rollout_utils.py:35,46) stays in the host wrapper.
 *   - Matrices are row-major fp32 (the reference is fp32 throughout; voltron/means/EWMA.py:37 even
 *     forces FloatTensor) or fp64 where a *_f64 twin exists.  `ld` = leading dimension in
 *     elements, `bs` = batch stride in elements.
 *   - "Working" matrices (A, Y) have dimension Np = volt_padded_n(N) (next multiple of 128); the
 *     padding is an identity block written by volt_prepare_*, so it contributes nothing to
 *     log-determinants, solves or traces.
 */
#ifndef VOLT_HIP_H
#define VOLT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VOLT_TILE 128
/* flag bits of the `flags` / `ws_flags` arguments below */
#define VOLT_WANT_GRAD 1          /* volt_mll_step_*: also produce the gradient outputs (the triangular inverse) */
#define VOLT_WS_INITIALISED 2     /* the workspace passed went through its *_workspace_init_f32 for this shape */
#define VOLT_REFINE_ALPHA 4       /* volt_mll_step_f32 with VOLT_WANT_GRAD: one step of iterative refinement of alpha */

/* ---- introspection ------------------------------------------------------------------------ */
int volt_abi_version(void);                 /* bumps when a signature changes */
const char* volt_source_hash(void);         /* hash of the sources this binary was built from (volt_amd/build.py) */
int volt_padded_n(int n);                   /* next multiple of VOLT_TILE */

/* ---- a1: CumTrapz  (voltron/kernels/VolKernel.py:4-10) -------------------------------------
 * V[b,i] = sum_{m<=i} w_m * y[b,m],  w = dx with first and last halved, dx = x[b,1]-x[b,0].
 * If `square` != 0, y = vol*vol is formed first, as VolatilityKernel.forward does (:28).
 * Bit-exact with the reference's CPU path: products in fp32, running sum in fp64, every prefix
 * rounded to fp32 (that is what torch.cumsum does on fp32 CPU tensors).
 * vol [B,N] (batch stride bs_vol), x [N] shared (bs_x = 0) or [B,N]; V [B,N] contiguous. */
int volt_cumtrapz_f32(const float* vol, int64_t bs_vol, const float* x, int64_t bs_x,
                      float* V, int B, int N, int square, void* stream);
int volt_cumtrapz_f64(const double* vol, int64_t bs_vol, const double* x, int64_t bs_x,
                      double* V, int B, int N, int square, void* stream);

/* ---- a2: VolatilityKernel.forward gather  (voltron/kernels/VolKernel.py:30-33) --------------
 * K[b,i,j] = V[b, min(i,j)], full square, row-major.  Replaces arange/meshgrid/minimum (int64
 * [N,N] index rebuilt on the CPU every call) + advanced-index gather.  HBM-write bound. */
int volt_fill_f32(const float* V, float* K, int B, int N, int64_t ldk, int64_t bsk, void* stream);
int volt_fill_f64(const double* V, double* K, int B, int N, int64_t ldk, int64_t bsk, void* stream);

/* ---- a4: EWMA  (voltron/means/EWMA.py:20-37) --------------------------------------------------
 * out[b,t] = sum_{j<k} w[j] * padded[b,t+j], padded = k copies of y[b,0] then y[b,:]; t = 0..N, so
 * out is [B,N+1]: out[:, :-1] is the train mean, out[:, -1] the one-step-ahead mean.  The k taps
 * `w` (device, fp32) are computed by the host exactly as the reference does (:21-24).  Replaces
 * conv1d + the forced .type(torch.FloatTensor) host round trip (:37). */
int volt_ewma_f32(const float* y, int64_t bs_y, const float* w, int k, float* out, int B, int N,
                  void* stream);

/* ---- a5: GaussianLikelihood "K + sigma^2 I"  (gpytorch, called from train_utils.py:249) ------
 * A[b] = tril-tiles(K[b]) + sigma2[b] I, padded to Np with an identity block.  sigma2 may be NULL
 * (adds nothing: rollouts factor raw K, rollout_utils.py:35) and `jitter` is added on top of it
 * (psd_safe_cholesky retry).  A [B,Np,Np] with ld = Np. */
int volt_prepare_f32(const float* K, int64_t ldk, int64_t bsk, const float* sigma2, float jitter,
                     float* A, int B, int N, void* stream);

/* ---- a5/a6: batched Cholesky  (torch.linalg.cholesky via gpytorch; psd_safe_cholesky at
 * rollout_utils.py:35, VoltronGP.py:83, VoltMagpie.py:87) ------------------------------------
 * In-place lower Cholesky of A [B,Np,Np] (left-looking, 128-wide panels, fp32 MFMA; ONE launch per block
 * column: diagonal tile, panel tiles (update + solve), look-ahead).  Winv [B, Np/128, 128, 128] receives the
 * inverses of the diagonal blocks (used by the solves; while the call runs, the first word of each block is also
 * the hand-off flag between the diagonal workgroup and the panel tiles of its launch).  info [B] int32: 0, the
 * 1-based index of the first non-positive / NaN pivot, or INT_MIN for an internal hand-off time-out (never
 * expected).  Diagonal tiles come back with their strict upper triangle zeroed; tiles above the diagonal are
 * never touched. */
int volt_potrf_f32(float* A, float* Winv, int* info, int B, int Np, void* stream);
/* The same with caller scratch (256-byte aligned, volt_potrf_workspace_bytes(B, Np) bytes; 0 for B >= 32): fewer than 32
 * matrices do not fill the chip with whole tiles, and with the scratch the long products of a launch are cut into
 * K-slices (csrc/chol.hip: split-K below 3 matrices, the balanced schedule of csrc/sched.h for the late block columns
 * of 3..31) -- one 4096^2 matrix 4.4 -> 1.3 ms.  ws == NULL is volt_potrf_f32. */
size_t volt_potrf_workspace_bytes(int B, int Np);
/* Once per scratch buffer (and again should the caller have overwritten it): copies the launch-schedule table for
 * (B, Np) into its table region, asynchronously on `stream`.  Optional: the factorisation follows the table only when
 * the caller passes ws_flags = VOLT_WS_INITIALISED (and then checks its header on the device: info = INT_MIN + 1 for
 * scratch that does not hold it); ws_flags = 0 runs the table-free schedules (3 .. 64 matrices: 2-25 % slower in the
 * late block columns). */
int volt_potrf_workspace_init_f32(void* ws, size_t ws_bytes, int B, int Np, void* stream);
int volt_potrf_ws_f32(float* A, float* Winv, int* info, int B, int Np, void* ws, size_t ws_bytes, int ws_flags, void* stream);
/* volt_prepare_f32 + volt_potrf_ws_f32 in one: the factor of K + (sigma2 + jitter) I (K [B,N,N], row stride ldk, batch
 * stride bsk, lower triangle read; sigma2 [B] or NULL) lands in A [B,Np,Np], Np = volt_padded_n(N), without a copy-in
 * pass -- the tiles are read from K by the workgroups that update them (what torch.linalg.cholesky(K + s2 I) is to the
 * reference: gpytorch's MLL, psd_safe_cholesky at rollout_utils.py:35).  ws as for volt_potrf_ws_f32 (may be NULL). */
int volt_potrf_k_f32(const float* K, int64_t ldk, int64_t bsk, const float* sigma2, float jitter, float* A, float* Winv,
                     int* info, int B, int N, void* ws, size_t ws_bytes, int ws_flags, void* stream);

/* fp64 twins on v_mfma_f64_16x16x4_f64 (the reference keeps the caller's dtype, VolKernel.py:28-33; the
 * noise-free train block of rollout_utils.py:35 has condition number 1e6 (N = 400) .. 1e8 (N = 4096), beyond
 * fp32).  Same layout and semantics as the fp32 pair; sigma2 [B] and jitter are doubles. */
int volt_prepare_f64(const double* K, int64_t ldk, int64_t bsk, const double* sigma2, double jitter,
                     double* A, int B, int N, void* stream);
int volt_potrf_f64(double* A, double* Winv, int* info, int B, int Np, void* stream);
/* The same with caller scratch (256-byte aligned, volt_potrf_workspace_bytes_f64(B, Np) bytes -- a few KB of progress words,
 * 0 where the shape does not use them; nothing to initialise): small batches of long series run the whole factorisation as
 * ONE launch (csrc/batch64_step.hip) instead of three launches per block column.  ws == NULL is volt_potrf_f64. */
size_t volt_potrf_workspace_bytes_f64(int B, int Np);
int volt_potrf_ws_f64(double* A, double* Winv, int* info, int B, int Np, void* ws, size_t ws_bytes, void* stream);
/* volt_prepare_f64 + volt_potrf_ws_f64 in one, the fp64 twin of volt_potrf_k_f32: the factor of K + (sigma2 + jitter) I with K
 * [B,N,N] read in place (row stride ldk, batch stride bsk; only the lower triangle) -- where the shape runs as one launch the
 * tiles come straight from K and A [B,Np,Np] only receives L (no copy-in pass; the strictly upper tiles of A are then not
 * written).  ws as for volt_potrf_ws_f64 (may be NULL: prepare + the launch-per-column schedule). */
int volt_potrf_k_f64(const double* K, int64_t ldk, int64_t bsk, const double* sigma2, double jitter, double* A, double* Winv,
                     int* info, int B, int N, void* ws, size_t ws_bytes, void* stream);

/* ---- a5/a6: triangular solves with one right-hand side  (torch.cholesky_solve at
 * rollout_utils.py:36,44; gpytorch inv_quad) ------------------------------------------------
 * rhs/out [B,Np] contiguous (pad with zeros).  `scratch` [B,Np] elements.  lower: out = L^-1 rhs;
 * lower_t: out = L^-T rhs.  rhs and out may alias.  ONE launch per solve: one workgroup per 128-block,
 * chained through release/acquire flags kept in `scratch`.  A hand-off that does not arrive within ~3 s of WALL CLOCK
 * (s_memrealtime; never expected -- only a bug or a wedged device gets there) turns the affected blocks into NaN
 * instead of hanging AND sets the error word: after the call, ((const int*)scratch)[1] != 0 says the solve timed out. */
int volt_trsv_lower_f32(const float* A, const float* Winv, const float* rhs, float* out,
                        float* scratch, int B, int Np, void* stream);
int volt_trsv_lower_t_f32(const float* A, const float* Winv, const float* rhs, float* out,
                          float* scratch, int B, int Np, void* stream);
int volt_trsv_lower_f64(const double* A, const double* Winv, const double* rhs, double* out,
                        double* scratch, int B, int Np, void* stream);
int volt_trsv_lower_t_f64(const double* A, const double* Winv, const double* rhs, double* out,
                          double* scratch, int B, int Np, void* stream);

/* ---- a5: triangular inverse for the noise gradient  (replaces autograd cholesky_backward) ---
 * Y = L^-T (upper triangular, row-major [B,Np,Np]); tr(K_s^-1) = ||Y||_F^2.  The fp64 twin uses the tiles BELOW the
 * diagonal of Y as scratch: only the upper triangle (diagonal included) is meaningful on return. */
int volt_trtri_f32(const float* A, const float* Winv, float* Y, int B, int Np, void* stream);
int volt_trtri_f64(const double* A, const double* Winv, double* Y, int B, int Np, void* stream);
/* The fp64 inverse with caller scratch (round 6): ws of volt_trtri_workspace_bytes_f64(B, Np) bytes (a few KB of progress words,
 * nothing to initialise; 0 = this shape has no one-launch schedule) lets the whole inverse run as ONE launch -- the rows' tiles
 * chase each other inside it (csrc/batch64_step.hip) instead of two launches per block row: 8 x 4096 4.8 -> 3.4 ms.  ws == NULL,
 * or a shape beyond the gate: exactly volt_trtri_f64.  Same result to rounding (different summation order of a tile's K blocks:
 * none -- the tiles' arithmetic is the launch-per-row kernels'). */
size_t volt_trtri_workspace_bytes_f64(int B, int Np);
int volt_trtri_ws_f64(const double* A, const double* Winv, double* Y, int B, int Np, void* ws, size_t ws_bytes, void* stream);

/* ---- a7/a8: sequential posterior rollouts  (voltron/rollout_utils.py:57-93 + :6-53) ----------
 * Bordered-Cholesky engine: the shared train block of every sample's matrix enters through two scalars per series,
 * rho = u'K^-1u and tau = u'K^-1 r_tr (the host gets them in closed form or from the fp64 factorisation,
 * volt_amd/rollout_engine.py); this launch walks all H horizon steps for every sample path: the sample's bordered
 * factor grown row by row, EWMA-family mean of the appended point (mean_mode 0 ewma / 1 dewma / 2 tewma /
 * 3 meanrevert, EWMA.py; 4 = a mean that depends on x alone -- constant, linear, log-linear, as in
 * experiments/weather/GPGenerator.py:68-82 --: hist_e1 then holds its values at the H test points, [G,H], hist_y is
 * ignored but must be valid with k = 1), optional mean reversion (:41-42), jitter ladder of psd_safe_cholesky(pred_cov, jitter) (:46).
 * G series x S samples x H steps, H <= VOLT_ROLLOUT_MAX_H.
 *   scratch == NULL  append-only solve: for the volatility kernel the right-hand side prefix and the stored rows
 *                    are step-invariant, so w_s only grows by one entry per step; nothing is stored or re-read.
 *   scratch != NULL  (volt_rollout_scratch_bytes(G,S,H) bytes) full forward substitution against the stored packed
 *                    rows at every step -- bitwise the same paths, H^3/6 * 4 B streamed per path; the cross-check.
 * hist_* [G,k] are the last k values of the (padded) train series / its EMA / EMA(EMA); acc0 [G] the CumTrapz running
 * sum through the last train point; rho, tau, acc0 are fp64 (the kernel works with acc - rho, which fp32 operands
 * would cancel away).  pred_vol, z, samples [G,S,H]; info [G,S]: 0; +step (1-based) of the first non-positive pivot
 * of the per-sample factor (a local jitter was applied); -step if the predictive variance stayed <= 0 after the
 * jitter ladder (the reference's psd_safe_cholesky raises NotPSDError there). */
#define VOLT_ROLLOUT_MAX_H 1024
size_t volt_rollout_scratch_bytes(int G, int S, int H);
int volt_rollout_bordered_f32(const double* rho, const double* tau, const double* acc0, const float* dx,
                              const float* hist_y, const float* hist_e1, const float* hist_e2,
                              const float* ema_prev, const float* mr_latent, const float* latent,
                              const float* w, const float* pred_vol, const float* z, float* samples,
                              float* scratch, int* info, int G, int S, int H, int k, int mean_mode,
                              int use_theta, float theta, float mr_theta, float jitter, void* stream);

/* nonvol_rollouts (voltron/rollout_utils.py:95-115): rollouts of a GP whose kernel is the same for every sample.
 * e [G,S,H] is the GP part of every path (posterior mean offset + correlated noise, computed by the caller from one
 * shared factorisation: e = c + M z, see volt_amd/rollout_engine.py); this adds the moving-average mean of the
 * sample's own stacked series, sequentially in the horizon: samples[.., idx] = mean_s(idx) + e[.., idx].
 * hist_* [G,k], ema_prev / mr_latent [G], w [k] and mean_mode as in volt_rollout_bordered_f32. */
int volt_rollout_shared_f32(const float* hist_y, const float* hist_e1, const float* hist_e2, const float* ema_prev,
                            const float* mr_latent, const float* w, const float* e, float* samples, int G, int S,
                            int H, int k, int mean_mode, float mr_theta, void* stream);

/* One-launch Adam step over a list of fp32 parameter tensors (train_utils.py:43,100,166,238,291 build
 * torch.optim.Adam(lr=0.1); same arithmetic, amsgrad off, no weight decay).  slots: DEVICE array of nslots records
 * {float* p; float* m; float* v; int64 end} (end = one past the tensor's last element in the flattened index space,
 * ascending); total = the last end; grad [total]: the gradients flattened in slot order.  state: 2 DEVICE ints {step
 * count t, arrival ticket}, zero before the first step; t is read and advanced on the device, so the launch can be
 * replayed from a graph. */
int volt_adam_step_f32(const void* slots, int nslots, long long total, const float* grad, float lr, float beta1, float beta2,
                       float eps, int* state, void* stream);

/* ---- a5: MLL + gradient  (ExactMarginalLogLikelihood + loss.backward(), train_utils.py:249-250)
 * One "step" of SURVEY 8(d) with K resident:
 *     A = K + sigma2 I -> potrf -> Y = L^-T -> z = Y'r, alpha = Y z
 *     out[b,0] = mll   = -1/2 (z'z + logdet + N log 2pi) / N
 *     out[b,1] = d mll / d sigma2 = 1/2 (alpha'alpha - tr K_s^-1) / N
 *     out[b,2] = z'z   out[b,3] = logdet   out[b,4] = tr K_s^-1   out[b,5] = alpha'alpha
 *     alpha[b,:] (= K_s^-1 r;  d mll / d mean = alpha / N)
 * resid [B,N] = y - mean(x).  `flags`: VOLT_WANT_GRAD -- without it the triangular inverse is skipped (forward
 * solve instead) and out[b,1], out[b,4], out[b,5] and alpha are not written; VOLT_WS_INITIALISED -- see below;
 * VOLT_REFINE_ALPHA (with VOLT_WANT_GRAD, opt-in) -- alpha <- alpha + K_s^-1 (r - K_s alpha) with the residual formed
 * against the caller's K in fp64 accumulation (BOTH triangles of K are read) and the correction solved through the
 * fp32 factor; out[b,0], out[b,1], out[b,2], out[b,5] are recomputed from the refined alpha and out[b,7] = 1.  Takes
 * alpha from the fp32 floor (cond * eps: 1e-5 .. 4e-5 of its max at sigma^2 = 1e-4, N = 4096, what the reference's own
 * fp32 path has) to < 1e-6 for one more pass over K and two triangular solves (+8 % of a step at 64 x 4096).
 * workspace: volt_mll_workspace_bytes(B,N,want_grad) bytes, 256-byte aligned. */
size_t volt_mll_workspace_bytes(int B, int N, int want_grad);
/* Once per workspace (and again should the caller have overwritten it), asynchronously on `stream`: copies the
 * launch-schedule table for this shape into the workspace (3 .. 31 series), and for short series (want_grad = 1, N <= 1024:
 * up to 40 series of N <= 512, 16 of N = 1024, 64 of N <= 256) writes the state of the ONE-LAUNCH step: a header, a step counter and the
 * flags its workgroups hand tiles on with (for ONE series of 1025 .. 4096 points also the list of its pieces, and the
 * workspace holds the slabs of their K-slices) -- volt_mll_step_f32 then enqueues one kernel for the whole step instead of
 * eleven (the counter lives on the device, so the launch replays from a hipGraph as it is).  Optional, like
 * volt_potrf_workspace_init_f32: the step uses what the init wrote only when the caller passes VOLT_WS_INITIALISED in
 * `flags` -- the library keeps no record of workspaces (rounds 2-3 recognised them by address) -- and every launch that
 * follows a table or the state checks the header there first: a region that does not hold what the init wrote (never
 * initialised, overwritten, or freed and handed out again) is reported (info = INT_MIN + 1), never followed.  Without the
 * flag the step runs the launch-per-column path.
 * (The workspace of volt_gpcv_step_f32 begins with an MLL workspace: same call, want_grad = 1.) */
int volt_mll_workspace_init_f32(void* workspace, int B, int N, int want_grad, void* stream);
int volt_mll_step_f32(const float* K, int64_t ldk, int64_t bsk, const float* resid,
                      const float* sigma2, float jitter, float* out /*[B,8]*/, float* alpha /*[B,N]*/,
                      int* info, void* workspace, int B, int N, int flags, void* stream);

/* The same step in double precision for double-precision models (the reference keeps the caller's dtype,
 * voltron/kernels/VolKernel.py:28-33; gpytorch computes log_prob in it): K, resid, sigma2, out [B,8], alpha [B,N] are
 * doubles; volt_potrf_f64 / volt_trsv_*_f64 / volt_trtri_f64 underneath.  Agreement with an fp64 LAPACK evaluation of
 * the same formulas: 1e-10 relative at N <= 1024, 1e-8 at N = 4096 (tests). */
size_t volt_mll_workspace_bytes_f64(int B, int N, int want_grad);
int volt_mll_step_f64(const double* K, int64_t ldk, int64_t bsk, const double* resid,
                      const double* sigma2, double jitter, double* out /*[B,8]*/, double* alpha /*[B,N]*/,
                      int* info, void* workspace, int B, int N, int want_grad, void* stream);

/* Dense gradient of the MLL wrt the covariance, for kernels whose parameters enter K elementwise (the fractional
 * Brownian-motion prior of the vol forecaster, voltron/models/BMGP.py:15-16 + kernels/FBMKernel.py:38-59):
 *     grad_K[b] = d mll_b / d K_b = 1/2 (alpha alpha' - K_s^-1) / N,   K_s^-1 = Y Y' on the structured GEMM.
 * Call after volt_mll_step_f32(want_grad = 1) on the same workspace; scratch [B, Np, Np] floats, 16-byte aligned. */
int volt_mll_grad_k_f32(void* mll_workspace, const float* alpha, float* scratch, float* grad_K, int B, int N,
                        void* stream);

/* ---- f4: GPCV volatility extraction (LearnGPCV, voltron/train_utils.py:15-67) --------------------
 * Structured batched GEMM on the fp32 MFMA core, C = alpha A B^T + beta C, row-major, K contiguous in
 * both operands.  uplo_* : 0 dense, 1 lower, 2 upper, at 128-tile granularity: for A (B) it restricts the
 * k-blocks read for a row-block, for C it selects the tiles written.  M, N, K multiples of 128;
 * lda/ldb multiples of 4, A and B 16-byte aligned. */
int volt_gemm_nt_f32(const float* A, int64_t lda, int64_t bsa, int uplo_a, const float* B, int64_t ldb,
                     int64_t bsb, int uplo_b, float* C, int64_t ldc, int64_t bsc, int uplo_c, float alpha,
                     float beta, int batch, int M, int N, int K, void* stream);

/* One ELBO + gradient evaluation of the variational GP whose inducing points are its inputs
 * (single_task_variational_gp.py:69-122 with use_whitened_var_strat=False; likelihood
 * volatility_likelihood.py:42-50, "exp" parameterisation; VariationalELBO call train_utils.py:44,51-54):
 *     q(u) = N(m, Lq Lq'),  prior N(mu, K + jitter I),
 *     ell = sum_i sum_k w_k log N(y_i; 0, max(exp(m_i + sqrt(2 var_i) x_k), min_scale)),  var_i = max(sum_j Lq_ij^2, min_var)
 *     KL  = 1/2 (tr(K^-1 S) + r'K^-1 r - N + logdet K - logdet S),   r = m - mu  (passed in as `resid`)
 * K [B,N,N] (ldk, bsk) without jitter; m, resid, y [B,N]; Lq [B,N,N] contiguous (upper triangle ignored);
 * gh_x, gh_w [Q] Gauss-Hermite nodes and weights / sqrt(pi).
 *     F   = w_ell ell - w_kl KL      (VariationalELBO: w_ell = 1/N, w_kl = beta/num_data)
 *     out[b,0..11] = ell, KL, r'K^-1 r, logdet K, logdet S, tr(K^-1 S), tr K^-1, |K^-1 Lq|_F^2, |K^-1 r|^2,
 *                    F, jitter, 0
 *     grad_m, grad_mu [B,N], grad_Lq [B,N,N]: gradients of F;  grad_K [B,N,N] (nullable) = dF/dK.
 * info[b] LAPACK-style for the factorisation of K + jitter I.  workspace: volt_gpcv_workspace_bytes. */
size_t volt_gpcv_workspace_bytes(int B, int N, int want_dk);
int volt_gpcv_step_f32(const float* K, int64_t ldk, int64_t bsk, float jitter, const float* resid, const float* m,
                       const float* Lq, const float* y, const float* gh_x, const float* gh_w, int Q, float min_var,
                       float min_scale, float w_ell, float w_kl, float* out, float* grad_m, float* grad_mu,
                       float* grad_Lq, float* grad_K, int* info, void* workspace, int B, int N, int ws_flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VOLT_HIP_H */
