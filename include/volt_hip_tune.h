/* volt_hip_tune.h -- measurement and tuning hooks of libvolt_hip.so.  NOT part of the drop-in boundary
 * (include/volt_hip.h): nothing on the product path calls these; bench.py's roofline leg, the scripts under scripts/
 * and tests/test_sched_host.py do.  They live in the same shared library so that what they time is the shipped code.
 *
 * Environment knobs.  The schedule defaults below are compiled in and were measured on MI355X; a
 * deployment reads NO environment variable.  Only a process started with VOLT_TUNE=1 (the experiment scripts) may
 * override them, through ONE table read once per process (csrc/chol.hip `tunables()`, csrc/chol64.hip `tune_int`):
 *   VOLT_GROUPS            stream groups for batches >= 16 (2)        VOLT_SPLITK_TARGET  workgroups per split launch (512)
 *   VOLT_SPLITK_MINL       shortest K-slice in blocks (2)            VOLT_SPLITK_MAXS    most slices per tile (8)
 *   VOLT_SPLITK_GROUPS / _MAXB   two split groups for 10 <= B < 22   VOLT_SCHED          balanced schedule on/off (1)
 *   VOLT_SCHED_MINB / _MAXB / _MAXB_POTRF   batch range of the balanced schedule (3 / 31 / 64)
 *   VOLT_SCHED_G / _S / _FRAC / _GROUPS / _KMIN   its slots (256), slices per tile (4), cut threshold (0.6), groups (2),
 *                          first scheduled block column (by batch size)
 *   VOLT_LONG / _NMIN / _FIRST / _EMIN / _PAD   ONE series as one launch: on (1), block columns above which it
 *                          takes a series (2), blocks in the slice next to the tile (0 = 3 up to 24 block columns, 4 above),
 *                          shortest sliced early part (> 1 block), one workgroup per CU (1)
 *   VOLT_LONG_SPLIT        its spine as two workgroups: S(g) solves tile (g,g-1) and hands its slabs on, R(g) takes the
 *                          rank-32 updates of the diagonal tile and diagonal block g (1; 0: one workgroup, +4 .. 13 %)
 *   VOLT_LONG_XCD          one long series: spines on one XCD, their slab hand-offs through its L2 (0; measured -1.7 %
 *                          with the one-workgroup spine, within noise with the split one)
 *   VOLT_EXTRA_FLAGS       (build time, volt_amd/build.py) extra hipcc flags for same-box A/B builds (scripts/ab_flags.sh)
 *   VOLT_PLAIN_SPREAD / VOLT_SPLIT_SPREAD   plain / all-split launches of up to this many workgroups run one workgroup
 *                          per CU (320 / 700)
 *   VOLT_BATCH / _ORDER / _LOCAL   the whole batched step in ONE launch (csrc/batch_step.hip): 0 off, 1 where measured faster
 *                          (default), 2 wherever it can run, 3 also ahead of the short- / long-series one-launch steps; order of
 *                          a column's tiles in the list (-1: by shape -- windows of 32 positions x 16 matrices from 48 x 3584 on, else
 *                          matrix innermost; VOLT_BATCH_LAD: look-ahead tiles listed this many columns early, 0: measured no gain);
 *                          hand-offs through the XCD's L2 when the batch is a multiple of 8 (1)
 *   VOLT_BATCH_PULLERS     the one-launch steps' grid: this many times the resident workgroups, pulling pieces by ticket (1)
 *   VOLT_BATCH_XSKEW / _XDROP   tests: the XCDs' queues shifted against the XCC ids / the pullers of the XCDs in the bit mask
 *                          leave at once (their queues are adopted): tests/test_gpu_topology.py
 *   VOLT_ROLLOUT_LANE      rollouts: one lane per path (1; 0: the wave-per-path engine everywhere)
 *   VOLT_F64_LOOKAHEAD     fp64 factorisation: look-ahead depth of the chain / bulk multi-stream schedule, 0 = one stream,
 *                          1 = one column, 2 = two columns (2 from 6 matrices on, else 1)           (read in csrc/chol64.hip)
 *   VOLT_F64_TRTRI_LOOKAHEAD  fp64 inverse: one-row look-ahead on its own stream, 0 / 1 (on up to 4 matrices)
 *   VOLT_F64_SPREAD        fp64: launches of up to this many workgroups run one workgroup per CU (512)
 *   VOLT_F64_SPLIT_TARGET  fp64: workgroups per K-sliced launch (512)
 * None of them is read by the product's Python; a deployment sets none.
 */
#ifndef VOLT_HIP_TUNE_H
#define VOLT_HIP_TUNE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement only (bench.py's roofline leg) -------------------------------------------------
 * Runs exactly the gradient step (the MLL entry point of volt_hip.h with want_grad = 1) ON THAT STEP'S OWN WORKSPACE AND
 * OUTPUTS -- same buffers, the fused z / Frobenius reductions in the trtri epilogues, the split-K scratch of the
 * small-batch schedules, the O(N^2) tail (sum_zpart, y_times_z, scalars) each group runs on its stream beside the other
 * groups' launches -- in the schedule the step uses (`groups` = 0: the library's default; > 0: that many stream groups,
 * 1 = lockstep, whole batch per launch on one stream), with every factorisation launch bracketed by HIP events on the
 * stream it is launched on.  Synchronises, and writes to HOST arrays, per kernel class, the summed launch durations,
 * the length of the UNION of the launch intervals (what the class occupied of the wall clock when launches of several
 * groups overlap) and the launch counts:
 *   [0] factor_step_kernel with a factorisation part (diagonal tile + look-ahead + panel tiles (update + solve) +
 *       trtri row k-1 in one grid, k = 0 .. n-1)          [1] factor_step_kernel carrying only the last trtri row.
 * out [B,8], alpha [B,N], workspace: as for the MLL step with want_grad = 1. */
int volt_profile_step_f32(const float* K, int64_t ldk, int64_t bsk, const float* resid, const float* sigma2,
                          float* out, float* alpha, void* workspace, int* info, int B, int N, int groups, void* stream,
                          float* ms_sum_host /*[2]*/, float* ms_union_host /*[2]*/, int* launches_host /*[2]*/,
                          float* per_launch_host /* NULL, or [every launch, in enqueue order] durations in ms */);

/* Tuning hook (scripts/tune_gemm.py): launches the panel tiles (update + solve, var 0) or the 2x2-wave diagonal
 * update (var 1) of block column k `reps` times on an already factored A; results are garbage, only the timing
 * matters. */
int volt_tune_update_f32(float* A, const float* Winv, int* info, int B, int Np, int k, int var, int reps, void* stream);
/* Host only (no GPU): the balanced schedule of launch k (k == n: the trailing trtri launch) of a factorisation of B
 * matrices with n block columns, as volt_potrf / the MLL step build it for mid-size batches (csrc/sched.h).  items
 * [max_items][4] int32 in grid order (kind | b << 3; block index; sl | nsl << 8 | tile << 16; b0 | b1 << 16), loads [G]
 * the load of each of G slots under greedy list scheduling of that order, in K-block units.  Returns the item count,
 * -1 bad argument, -2 max_items too small. */
int volt_sched_describe(int B, int n, int has_y, int k, int G, int S, float frac, int* items, int max_items, float* loads);
/* The diagonal-block kernel alone on block column k of B (unfactored) matrices, with s_memtime stamps of its
 * phases: stamps [B,32] int64 (load, factor32 x4 with panel / trailing updates, L out, inverse, W out, publish). */
int volt_tune_diag_f32(float* A, float* Winv, int* info, int B, int Np, int k, long long* stamps, void* stream);

/* The fp64 diagonal-block kernel alone on block column k, with s_memtime stamps of its phases: stamps [B,32] int64
 * (0 start, 1 loaded, per 32-wide sub-block kb: 2+4kb factored, 3+4kb L out + inverted, 4+4kb panel, 5+4kb trailing;
 * 18 L out, 19 W computed, 20 W out). */
int volt_tune_diag_f64(double* A, double* Winv, int* info, int B, int Np, int k, long long* stamps, void* stream);

/* The one-launch step for short series (csrc/chol.hip, small_step_kernel): while `stamps` (device, 16 int64 per workgroup
 * of the launch, one per piece) is set, every workgroup of the following steps records s_memrealtime (100 MHz) at
 * 0 entry, 1 first wait over, 2 second wait over, 3 work done, 4 published, 5 tail done / exit; the spine also 6..9 slab
 * flag j seen, 10 last slab solved, 11 all waves there, 12 last rank-32 update done, 13 pivot image written, 14 tile out.  NULL switches it off. */
int volt_tune_small_stamps(long long* stamps);

/* Host only: the piece list of the one-launch step for ONE LONG series of n block columns (csrc/long_sched.h): early
 * parts longer than `emin` K blocks cut into slices that grow geometrically away from the tile, the last one `first`
 * blocks (first = 0 / emin = -1: the values the step itself uses; with the spine split -- the default -- S(g) is followed
 * by R(g), kind 9, which owns diagonal block g).  items: up to max_items records of 4 ints {kind | a << 8 | b << 16, slices or b0 | b1 << 8, slab slot, slice
 * counter}; nslabs / ncnt (optional): slab slots and slice counters.  Returns the number of pieces (= workgroups). */
int volt_long_describe(int n, int first, int emin, int* items, int max_items, int* nslabs, int* ncnt);

/* Host only: the piece list of the one-launch BATCHED step (csrc/batch_sched.h, csrc/batch_step.hip) for B matrices of n
 * block columns, in grid order: items [max_items][4] int32 {kind | b << 3, row, col, 0}, kind 0 diagonal tile D(k = row),
 * 1 look-ahead for tile (row+1,row+1), 2 panel tile (row, col), 3 tile (row, col) of the inverse, 4 its diagonal tile.
 * order: bit 0: 0 positions in column order with the matrix innermost, 1 matrix-group-major inside a block column; bits
 * 1.. : how many block columns early the look-ahead tiles are listed (VOLT_BATCH_LAD); 1000 + w: the windowed order w of
 * batch_sched.h (2 + 4 log2(window) + 32 (groups of 8 matrices side by side - 1)); -1: the order the step itself picks for
 * this shape.  items may be NULL (count only).  Returns the number of pieces (= tickets of the launch: since round 6 a grid
 * of resident pullers takes them by ticket), -1 bad argument, -2 max_items too small. */
int volt_batch_describe(int B, int n, int has_y, int order, int* items, int max_items);
/* The one-launch batched step: while `stamps` (device, 8 int64 per workgroup of the launch = per piece of the list) is set,
 * every workgroup of the following steps records [0] s_memrealtime (100 MHz) at entry, [1] at exit, [2] XCC_ID << 32 | HW_ID,
 * two-phase tiles also [3] / [4] entering / leaving the tile pipeline.  NULL switches it off. */
int volt_tune_batch_stamps(long long* stamps);
/* The fp64 one-launch step (csrc/batch64_step.hip).  Host only: its piece list in grid order, items [max_items][4] int32
 * {kind, row, col, matrix}: kind 0 diagonal block D(k = row), 1 panel tile (row, col), 2 diagonal tile of the inverse, 3 tile
 * (row, col) of the inverse.  items may be NULL (count only).  Returns the number of pieces, -1 bad argument, -2 too small. */
int volt_batch64_describe(int B, int n, int has_y, int* items, int max_items);
/* ... and its stamps, 8 int64 per workgroup: [0] entry, [1] exit, [2] XCC_ID << 32 | HW_ID, [3] behind the chased product,
 * [4] behind the wait for W, [5] behind the second product.  VOLT_BATCH64 = 0 / 1 / 2: off / where measured faster / wherever
 * it can run; VOLT_BATCH64_MAX: tiles per block column, B (n + 1), up to which it runs. */
int volt_tune_batch64_stamps(long long* stamps);
/* Host only: what the library takes the device to be and the schedule gates that follow (csrc/host.h).  out [7] int32:
 * CUs, XCDs, workgroup slots the balanced schedule plans for, plain / split launches up to this many workgroups run one
 * per CU, launches below this many workgroups run as one stream group, one-launch steps enabled (bit 0 short series, 1 one
 * long series, 2 batched).  Every gate was measured on the full MI355X (256 CUs in 8 XCDs); on any other device -- a CPX /
 * DPX partition, a reduced part -- the slot gates are scaled with the CU count and the one-launch steps are off.
 * VOLT_TUNE=1 VOLT_FAKE_CUS / VOLT_FAKE_XCCS plan as if for such a device (tests). */
int volt_topology_describe(int* out);

#ifdef __cplusplus
}
#endif
#endif /* VOLT_HIP_TUNE_H */
