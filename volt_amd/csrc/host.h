// Host-side internals shared by the translation units of libvolt_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// Schedule parameters: compiled-in defaults (measured on MI355X).  A deployment reads NO environment:
// only a process started with VOLT_TUNE=1 (the experiment scripts) may override them through the VOLT_* variables listed in
// include/volt_hip_tune.h -- read ONCE, here and in chol64.hip's tune_int, nowhere else.
struct Tunables {
    int groups = 2;                  // stream groups of a large batch (2 >= 4 > 8 with one launch per block column)
    int splitk_target = 512, splitk_minl = 2, splitk_maxs = 8, splitk_groups = 2, splitk_maxb = 22;
    int sched = 1, sched_minb = 3, sched_maxb = 31, sched_maxb_potrf = 64;
    int sched_g = 256, sched_s = 4, sched_groups = 2, sched_kmin = -1;
    float sched_frac = 0.6f;
    int small_nmax = 8, small_maxwg = 1150;    // the one-launch step: block columns it takes, and workgroups at most
    int plain_spread = 320, split_spread = 700;   // plain / all-split launches of up to this many workgroups run one workgroup per CU
    int long_split = 1;                      // long series: the spine split in two workgroups (substitution | rank-32 updates + diagonal block)
    int long_xcd = 0;                        // long series: spines on one XCD, their hand-offs through its L2 (long_sched.h)
    int long_on = 1, long_first = 0, long_emin = 1, long_pad = 1, long_nmin = 2;   // one long series in one launch: on/off, last slice's blocks (0: 3 up to 24 block columns, 4 above), shortest sliced early part, a CU per workgroup, block columns above which it takes one series
    int small_maxb = 40, small_maxb2 = 64;     // ... series at most (three or four block columns / one or two)
    int small_pad_maxb = 40;                   // ... up to this many series with a CU per workgroup (16 KB of LDS padding)
    // ---- topology (round 5).  Every gate above was measured on a full MI355X: 256 CUs in 8 XCDs.  The device is asked once
    // (multiprocessor count, hipDeviceAttributeNumberOfXccs); on anything else -- a partitioned (CPX / DPX) or otherwise
    // reduced device -- the slot-count gates are scaled with the CU count and the one-launch steps, whose tuning AND whose
    // one-XCD-per-matrix hand-offs assume the full chip, are switched off: such a device runs the launch-per-column schedules.
    int cus = 256, xccs = 8;
    int group_gate = 700;                      // launches of fewer workgroups than this run as one stream group
    int batch = 1;                             // the whole batched step in one launch (batch_step.hip): 0 off, 1 where measured faster, 2 wherever it can run
    int batch_order = -1;                      // ... order of a block column's tiles in its list: -1 by shape (batch_step.hip, batch_order_for), 0 matrix innermost, 1 matrix-major, 2 + 4 log2(window) + 32 (groups side by side - 1) windowed (batch_sched.h)
    int batch_lad = 0;                         // ... its look-ahead tiles listed this many block columns early (batch_sched.h).  Measured, ms/step at lad 0 / 1 / 2 / 3: 8 x 4096 3.64 / 3.65 / 3.68 / 3.69, 16 x 4096 5.94 / 5.95 / 6.00 / 6.13, 64 x 4096 21.95 / 21.98 / 21.99 / 22.01: no gain, off
    int batch_spread = 400;                    // ... with up to this many tiles per block column, B (n + 1), it runs ONE workgroup per CU.  ms/step two per CU -> one per CU at N = 4096: B = 2 1.90 -> 1.66, 4 2.57 -> 2.10, 8 3.64 -> 3.37, 12 4.80 -> 4.82, 16 5.89 -> 6.16, 24 8.51 -> 9.17; at N = 2048: B = 8 0.99 -> 0.82, 16 1.28 -> 1.11.  Shorter series cross over later (their tiles are shorter, the chain weighs more): + 9 tiles per block column short of 32 -- 24 x 2048 (408 tiles) 1.59 -> 1.48, 32 x 2048 (544) 1.840 -> 1.836, 64 x 2048 (1088) 3.26 -> 3.52; 32 x 1536 (416) 1.02 -> 0.92; 64 x 1024 (576) 0.70 -> 0.67, 40 x 1024 0.58 -> 0.48; 16 x 3072 (400) 2.92 -> 2.87, 24 x 3072 (600) 3.97 -> 4.21
    int batch64 = 1;                           // the fp64 factorisation / gradient step in one launch (batch64_step.hip): 0 off, 1 where measured faster, 2 wherever it can run
    int batch64_max = 3300;                    // ... up to this many tiles per block column, B (n + 1) (one workgroup per CU: the diagonal block's 133 KB image).  Measured against chol64.hip's schedules (profiles/r05/batch64_gate_sweep.txt, _large.txt): the factorisation 1.06 - 2.17 x at every shape tried, 1 .. 512 matrices of N = 512 .. 4096 (up to 3264 tiles); the gradient step 1.01 - 1.90 x (96 x 4096, 3168 tiles: 1.01; 96 x 3072, 2400: 1.05): its limit from 24 block columns on is batch64_max_step
    int batch64_max_step = 2500;
    int batch_local = 1;                       // ... batches that are a multiple of 8 hand their tiles on through the XCD's L2 (0: the agent-scope protocol everywhere)
    int batch_pullers = 1;                     // ... its grid: this many times the workgroups the chip holds at once, pulling pieces by ticket (0: one workgroup per piece -- they pull all the same)
    int long_pullers = 1;                      // one long series: this many times the resident workgroups pull its pieces by ticket (0: a workgroup per piece -- they take tickets all the same)
    int rollout_lane = 1;                      // rollouts: one lane per path where the mean's window fits the LDS ring (0: a wave per path everywhere -- the tests compare the two)
    int batch_xskew = 0, batch_xdrop = 0;      // ... tests only: the queues of the XCDs shifted by this many (the map is nobody's assumption); bit x set = the pullers on XCD x leave at once (an XCD a CU mask emptied: its queue is adopted)
};
// (the struct continues: what the device looks like, and what follows from it)
const Tunables& tunables();                    // chol.hip: read once per process (VOLT_TUNE=1 only)
