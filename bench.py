#!/usr/bin/env python3
"""bench.py -- MLL+grad steps/s at N=4096, batch=64 per GPU (BASELINE.json metric), MI355X.

    python bench.py [--gpus N --steps K --warmup W]

N > 1 runs one process per GPU over RCCL.  Under a launcher (torch.distributed.run sets WORLD_SIZE) this process is one
rank; without one, `python bench.py --gpus N` starts the N ranks ITSELF (re-exec under torch.distributed.run,
127.0.0.1 rendezvous) and fails loudly when fewer than N GPUs are visible -- it never prints an N-GPU line from
fewer ranks (`ranks_seen` in the JSON is an all-reduce of ones).

A *step* is one pass of the training-loop body of voltron/train_utils.py:243-254 over a batch of
independent series with K (train_cov) already resident: residual y - EWMA mean -> K + sigma^2 I
-> blocked Cholesky -> L^-T -> MLL and its gradient wrt raw_noise -> Adam update of raw_noise ->
(N>1) one RCCL all-reduce of the summed loss/grad scalars.  Series shard across ranks, no data-path
collective.  --scaling weak (default): 64 series per GPU; --scaling strong: 64 series in total, split
evenly over the ranks (BASELINE.json's metric reads "N=4096, batch=64; 1/2/4/8 GPU").  With N > 1 the OTHER mode is
timed right after the headline region and reported beside it (`other_scaling`).

Rank 0 prints ONE JSON line.  `value` = batch-of-64 steps per second summed over ranks, inputs
resident in HBM.  `roofline` is for the dominant kernel (per-launch HIP events in the timed schedule);
`cpu_baseline` times the torch-CPU restatement of the gpytorch path on a bounded sample (and real
gpytorch beside it where it is importable); `rollouts` times BASELINE config 5's per-GPU share.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SERIES = 4096          # BASELINE.json metric: N = 4096
BATCH = 64               # series per GPU
EWMA_K = 25              # VoltMagpie default k (voltron/models/VoltMagpie.py:17)
FP32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBS = 8000.0


def kernel_class_flops(B: int, Np: int):
    """ALGORITHMIC flops of the two launch classes over one factorisation + inverse of B matrices, counted so that the
    Cholesky parts sum to exactly Np^3/3 and the triangular inverse to Np^3/3 (work the kernels do on structural
    zeros -- the upper half of diagonal tiles, the zero half of triangular operands -- is NOT credited):
      panel tile (i,k), i>k: 2*128^3*k (update) + 128^3 (solve against the triangular W_k);  diagonal tile: 128^3*k
      (update) + 128^3/3 (factor) + 128^3/3 (inverse, credited to the trtri total);  trtri tile (i,j), i>j: 2*128^3*(i-j).
    Returns [step launches (block columns 0..n-1 with trtri rows 0..n-2 aboard), the trailing trtri row]."""
    n, c = Np // 128, float(128 ** 3)
    upd = sum((n - k - 1) * k * 2 * c + k * c for k in range(1, n))      # panel + diagonal updates
    chol_diag = n * c / 3
    trsm = sum(n - k - 1 for k in range(n)) * c
    assert abs(upd + chol_diag + trsm - Np ** 3 / 3) < 1e-6 * Np ** 3
    row = lambda i: sum(2 * c * (i - j) for j in range(i))               # trtri row i (off-diagonal tiles)
    tri_diag = n * c / 3
    assert abs(sum(row(i) for i in range(n)) + tri_diag - Np ** 3 / 3) < 1e-6 * Np ** 3
    step = upd + chol_diag + trsm + tri_diag + sum(row(i) for i in range(n - 1))
    return [B * step, B * row(n - 1)]


def launch_ranks(n_gpus: int, one_dev: bool) -> int:
    """`python bench.py --gpus N` without a launcher: one process per GPU, started here the way the driver starts them
    (torch.distributed.run, one node, 127.0.0.1 rendezvous).  Refuses when the box has fewer than N GPUs."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < n_gpus and not one_dev:
        print(f"bench.py: --gpus {n_gpus} needs {n_gpus} visible GPUs, this box has {ndev}; not printing an "
              f"{n_gpus}-GPU line from fewer ranks", file=sys.stderr)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)       # ~4.6 s of GPU time: long enough for external samplers
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --batch series per GPU; strong: --batch series in total, split over the ranks")
    ap.add_argument("--no-rollouts", action="store_true", help="skip the config-5 rollout leg")
    ap.add_argument("--n", "--series-len", dest="n", type=int, default=N_SERIES)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (the other BASELINE shapes)")
    ap.add_argument("--with-rollouts", action="store_true", help="run the rollout leg even under --no-aux-legs")
    ap.add_argument("--rollout-samples", type=int, default=10000)
    ap.add_argument("--rollout-horizon", type=int, default=256)
    ap.add_argument("--profile-only", type=int, default=0, metavar="N",
                    help="for `rocprofv3 --kernel-trace --stats`: after the set-up run EXACTLY N gradient steps through the "
                         "roofline hook (each bracketed by HIP events and synchronised), no warm-up, no other leg, and print "
                         "their mean kernel time: the trace's average duration of the step kernel over the same N launches "
                         "reproduces roofline.class_ms")
    ap.add_argument("--no-aux-legs", action="store_true",
                    help="skip the ms/Cholesky and forward-only legs (used for the rocprofv3 summaries in profiles/, so "
                         "that every factor_step_kernel<true> launch in the trace is a gradient-step launch)")
    args = ap.parse_args()

    one_dev = os.environ.get("VOLT_BENCH_ONE_DEVICE") == "1"     # dry-run hook (tests): every rank on device 0 over gloo
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, one_dev))               # start the N ranks ourselves; this process only waits
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    ndev = torch.cuda.device_count()
    if one_dev:
        local_rank = 0
    if ndev <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but only {ndev} GPU(s) are visible "
                         f"(--gpus {args.gpus}); refusing to share a device")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("VOLT_BENCH_BACKEND", "nccl")          # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from volt_amd import _lib, ops
    from volt_amd.synthetic import sde_batch

    ranks_seen = 1
    if dist is not None:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                                    # every rank that really runs adds one
        ranks_seen = int(ones.item())
        if ranks_seen != world:
            raise SystemExit(f"bench.py: {ranks_seen} ranks answered the all-reduce, {world} expected")

    n = args.n
    if args.batch % world:
        raise SystemExit(f"batch {args.batch} does not divide over {world} ranks (needed for the strong-scaling leg)")
    B_by_mode = {"weak": args.batch, "strong": args.batch // world}     # series per rank
    B = B_by_mode[args.scaling]
    Bmax = max(B_by_mode.values())
    x, F, vol = sde_batch(Bmax, n, seed=2019, first=rank * Bmax)        # this rank's shard of series
    xd, vold = torch.tensor(x, device=dev), torch.tensor(vol, device=dev)
    y_all = torch.log(torch.tensor(F[:, 1:], device=dev))

    # ---- fill (timed separately): K is built once per model (VoltMagpie.py:46)
    V = ops.cumtrapz(vold, xd, square=True)
    K_all = ops.fill(V)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fills = []
    for _ in range(7):
        e0.record()
        K_all = ops.fill(V)
        e1.record()
        torch.cuda.synchronize()
        fills.append(e0.elapsed_time(e1))
    fill_ms = float(np.median(fills))
    fill_gbs = Bmax * n * n * 4 / (fill_ms * 1e-3) / 1e9

    def make_step(Bs, Ks=None, ys=None, n_=None):
        """The training-loop body over Bs series (K resident): this rank's first Bs series of the headline shape, or the
        given (Ks, ys) of another shape (the `configs` block) -- the SAME body either way."""
        if Ks is None:
            Ks, ys, n_ = K_all[:Bs], y_all[:Bs], n
        raw = torch.full((Bs,), 1e-5, device=dev, requires_grad=True)    # train_utils.py:222
        opt_ = torch.optim.Adam([raw], lr=0.1)                             # train_utils.py:236-238
        ws_ = ops.MllWorkspace(Bs, n_, True, dev)
        red_ = torch.zeros(2, device=dev)

        def step():
            opt_.zero_grad(set_to_none=False)
            mean = ops.ewma(ys, EWMA_K)[..., :-1]                  # EWMAMean.forward (EWMA.py:46-54)
            resid = ys - mean
            with torch.no_grad():
                sigma2 = torch.nn.functional.softplus(raw) + 1e-4
            out, alpha, info = ops.mll_step(Ks, resid, sigma2, ws_, want_grad=True)
            with torch.no_grad():
                # loss = -mll ; d loss / d raw = -(d mll / d sigma2) * sigmoid(raw)
                raw.grad = -(out[:, 1] * torch.sigmoid(raw))
                red_[0] = -out[:, 0].sum()
                red_[1] = raw.grad.sum()
            if dist is not None:
                dist.all_reduce(red_)                              # the path's only collective
            opt_.step()
            return out, info
        return step, raw, ws_, red_

    def timed(step, warmup, steps):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronize; max over ranks."""
        for _ in range(warmup):
            step()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out, info = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt_ = time.perf_counter() - t0
        per_rank = [dt_]
        if dist is not None:
            tt = torch.tensor([dt_], device=dev, dtype=torch.float64)
            gathered = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(gathered, tt)
            per_rank = [float(g.item()) for g in gathered]
        return max(per_rank), per_rank, info

    step, raw_noise, ws, red = make_step(B)
    if args.profile_only > 0:
        profile_only_leg(args.profile_only, K_all[:B], y_all[:B], raw_noise, ws, n, B, dev)
        return
    dt, dt_ranks, info = timed(step, args.warmup, args.steps)
    bad = int((info != 0).sum().item())
    loss = float(red[0].item()) / (B * world)
    K, y = K_all[:B], y_all[:B]

    other = None
    if world > 1:                       # the other scaling mode, same steps, right behind the headline region
        om = "strong" if args.scaling == "weak" else "weak"
        Bo = B_by_mode[om]
        del ws
        step_o, _, ws_o, _ = make_step(Bo)
        dto, dto_ranks, info_o = timed(step_o, args.warmup, args.steps)
        other = {"scaling": om, "series_per_gpu": Bo, "series_total": Bo * world,
                 "value": round(Bo * world / args.batch * args.steps / dto, 4), "ms_per_step": round(dto / args.steps * 1e3, 3),
                 "ms_per_step_by_rank": [round(t / args.steps * 1e3, 3) for t in dto_ranks],
                 "not_pd": int((info_o != 0).sum().item())}
        del step_o, ws_o
        ws = ops.MllWorkspace(B, n, True, dev)

    # ---- roofline leg: per-launch HIP events (on the stream each launch goes to) around every launch of one
    # factor+inverse in the SAME schedule the timed steps use; per class the union of the launch intervals
    roof = None
    extra = {}
    if rank == 0:
        Np = ops.padded_n(n)
        L = _lib.lib()
        s2 = (torch.nn.functional.softplus(raw_noise.detach()) + 1e-4).contiguous()
        inf = torch.empty(B, dtype=torch.int32, device=dev)
        resid_p = (y - ops.ewma(y, EWMA_K)[..., :-1]).contiguous()

        def profile(groups, reps=4):
            ms_sum, ms_un, cnt = (ctypes.c_float * 2)(), (ctypes.c_float * 2)(), (ctypes.c_int * 2)()
            tot_s, tot_u = np.zeros(2), np.zeros(2)
            for rep in range(reps + 1):                      # the first pass (event creation, cold tables) is not counted
                # the step's own workspace: same buffers, fused reductions and scratch as the timed steps
                _lib.check(L.volt_profile_step_f32(K.data_ptr(), n, n * n, resid_p.data_ptr(), s2.data_ptr(), ws.out.data_ptr(),
                                                   ws.alpha.data_ptr(), ws.ptr, inf.data_ptr(), B, n, groups,
                                                   _lib.stream_ptr(), ms_sum, ms_un, cnt, None), "profile")
                if rep == 0:
                    continue
                tot_s += np.array(list(ms_sum))
                tot_u += np.array(list(ms_un))
            return tot_s / reps, tot_u / reps, list(cnt)

        names = ["factor_step_kernel<true>", "factor_step_kernel<false>(last trtri row)"]
        flops = kernel_class_flops(B, Np)
        sum_t, un_t, cnt = profile(0)                       # the schedule the timed steps ran in
        sum_1, un_1, cnt1 = profile(1)                      # launch per block column, lockstep: one stream, whole batch per launch
        one_launch = cnt[0] == 1 and cnt[1] == 0            # the one-launch batched step (csrc/batch_step.hip): factor AND inverse
        if one_launch:
            names_t, flops_t = [f"batch_step_kernel<true, {'true' if B % 8 == 0 else 'false'}>"], [flops[0] + flops[1]]
        else:
            names_t, flops_t = names, flops
        dom = 0 if one_launch else int(np.argmax(un_t))
        ach = flops_t[dom] / (un_t[dom] * 1e-3) / 1e12
        traffic, traffic_source = pmc_traffic(names_t[dom].split("(")[0], n, B)
        ngroups = 1 if one_launch else cnt[0] // max(1, cnt1[0])
        dom1 = int(np.argmax(un_1))
        roof = {"kernel": names_t[dom], "bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TF,
                "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TF, 4), "traffic": traffic, "traffic_source": traffic_source,
                "launches": int(cnt[dom]), "avg_launch_ms": round(float(sum_t[dom] / max(1, cnt[dom])), 4),
                "class_ms": round(float(un_t[dom]), 3),
                "measured_in": ("the timed schedule: ONE launch per step (factorisation + inverse + alpha's partial sums of all "
                                f"{B} series), HIP events around it on its stream; achieved = the step's algorithmic 2 N^3 / 3 flop "
                                "per series / that launch's duration (class_ms); `bench.py --profile-only N` under rocprofv3 "
                                "--kernel-trace --stats reproduces it as the kernel's average duration") if one_launch else
                               (f"the timed schedule ({ngroups} groups of {B // max(1, ngroups)} series on {ngroups} streams): HIP "
                                "events on each launch's own stream; achieved = algorithmic flops of the class / the UNION of "
                                "its launch intervals (class_ms); avg_launch_ms = mean duration of one group's launch while the "
                                "other groups' launches share the GPU (traffic is per whole-batch launch, lockstep PMC passes)"),
                "lockstep": {"what": "the launch-per-block-column schedule on one stream (round 4's kernel), for comparison",
                             "kernel": names[dom1], "class_ms": round(float(un_1[dom1]), 3), "launches": int(cnt1[dom1]),
                             "avg_launch_ms": round(float(sum_1[dom1] / max(1, cnt1[dom1])), 4),
                             "achieved": round(flops[dom1] / (un_1[dom1] * 1e-3) / 1e12, 2),
                             "frac": round(flops[dom1] / (un_1[dom1] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4)}}
        names, flops = names_t, flops_t
        extra = {"kernel_ms": {nm: round(float(t), 3) for nm, t in zip(names, un_t)},
                 "kernel_tflops": {nm: round(f / (t * 1e-3) / 1e12, 2) for nm, f, t in zip(names, flops, un_t)},
                 "factor_plus_inverse_ms": round(float(un_t.sum()), 3),
                 "fill_ms": round(fill_ms, 3), "fill_GBps": round(fill_gbs, 1),
                 "fill_frac_of_hbm_peak": round(fill_gbs / HBM_PEAK_GBS, 4)}
        if args.no_aux_legs:
            extra["ms_per_cholesky"] = None
        # ---- ms/Cholesky (the other half of BASELINE.json's metric): one batched factorisation of K + s2 I as
        # shipped (copy-in + blocked Cholesky, 4-stream schedule), and the forward-only MLL built on it
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        L_ = _lib.lib()
        legs = () if args.no_aux_legs else ("chol", "fwd")

        A = Winv = None
        if legs:
            A = torch.empty(B, Np, Np, device=dev)
            Winv = torch.empty(B, Np // 128, 128, 128, device=dev)
        pws_bytes = int(L_.volt_potrf_workspace_bytes(B, Np))   # scratch of the late-column schedule (what ops.potrf passes)
        pws = torch.empty(pws_bytes + 256, dtype=torch.uint8, device=dev) if pws_bytes else None
        pws_ptr = ((pws.data_ptr() + 255) // 256) * 256 if pws is not None else None
        if pws is not None:
            _lib.check(L_.volt_potrf_workspace_init_f32(pws_ptr, pws_bytes, B, Np, _lib.stream_ptr()), "potrf workspace init")

        def potrf_once():                                   # what ops.potrf calls: the factor of K + s2 I straight from K
            _lib.check(L_.volt_potrf_k_f32(K.data_ptr(), n, n * n, s2.data_ptr(), 0.0, A.data_ptr(), Winv.data_ptr(),
                                           inf.data_ptr(), B, n, pws_ptr, pws_bytes,
                                           _lib.WS_INITIALISED if pws_ptr else 0, _lib.stream_ptr()), "potrf")

        def fwd_once():
            _lib.check(L_.volt_mll_step_f32(K.data_ptr(), n, n * n, y.data_ptr(), s2.data_ptr(), 0.0, ws.out.data_ptr(),
                                            ws.alpha.data_ptr(), ws.info.data_ptr(), ws.ptr, B, n, 0, _lib.stream_ptr()),   # forward only: the workspace was sized and initialised for the gradient step, so no VOLT_WS_INITIALISED here
                       "mll fwd")
        res = {}
        for name, fn in (("chol", potrf_once), ("fwd", fwd_once)):
            if name not in legs:
                continue
            fn()
            torch.cuda.synchronize()
            c0.record()
            for _ in range(3):
                fn()
            c1.record()
            torch.cuda.synchronize()
            res[name] = c0.elapsed_time(c1) / 3
        f = None
        if legs:
            extra["ms_per_cholesky"] = round(res["chol"] / B, 4)
            extra["cholesky_tflops"] = round(B * Np ** 3 / 3 / (res["chol"] * 1e-3) / 1e12, 2)
            extra["ms_per_mll_forward"] = round(res["fwd"] / B, 4)
        extra["cholesky_note"] = ("ms_per_cholesky = wall time of one batched factorisation of K + s2 I (volt_potrf_k_f32: blocked "
                                  "Cholesky reading K in place, N^3/3 flop each) / 64; ms_per_mll_forward adds the forward solve "
                                  "and log-det")
        del A, Winv, f, pws

    # ---- rollouts leg: BASELINE config 5 (64 series x 10,000 paths x 256 steps over 8 GPUs).  N = 1: its per-GPU share,
    # 8 series.  N > 1: EVERY rank rolls out its 64 / N series (series shard first), the job's rate is the
    # series of all ranks / the slowest rank, and the final gather of the [S, H] samples -- the one place xGMI bandwidth
    # is exercised -- is timed on its own.
    roll = None
    if not args.no_rollouts and (args.with_rollouts or not args.no_aux_legs):
        G_roll = 8 if world == 1 else max(1, min(Bmax, 64 // world))
        roll = rollout_leg(x, F, vol, dev, n, G=G_roll, S=args.rollout_samples, H=args.rollout_horizon,
                           dist=dist, world=world, rank=rank)

    # ---- the other BASELINE shapes (rank 0, N=1): same loop body, same timing loop as the headline
    cfgs = None
    if rank == 0 and world == 1 and not args.no_aux_legs and not args.no_configs:
        del ws
        cfgs = configs_leg(make_step, timed, dev, K_all, y_all, n, args.batch, api_inputs=(x, F, vol))
        ws = ops.MllWorkspace(B, n, True, dev)

    # ---- API leg (rank 0, N=1): what a caller of the drop-in surface pays per iteration
    api = None
    if rank == 0 and world == 1 and not args.no_aux_legs:
        api = api_leg(x, F, vol, dev, n, B, dt / args.steps * 1e3)

    # ---- fp64 leg (rank 0): the double-precision twins of the step, at the 8-GPU share of the batch
    f64 = None
    if rank == 0 and not args.no_aux_legs:
        f64 = f64_leg(x, F, vol, dev, n, min(8, B))

    # ---- the (f) rows: GPCV's ELBO step and the vol model's step, at the reference's sizes and at 8 x 4096 (rank 0, N=1)
    nxt = None
    if rank == 0 and world == 1 and not args.no_aux_legs:
        try:
            nxt = next_rows_leg(dev)
        except Exception as e:
            nxt = {"error": repr(e)[:300]}

    # ---- CPU baseline leg (rank 0, N=1 only): torch-CPU restatement of the gpytorch path (+ real gpytorch if present)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_leg(K, y, ops.ewma(y, EWMA_K)[..., :-1], n, args.batch)

    if rank == 0:
        per_step_series = world * B
        steps_per_s = per_step_series / args.batch * args.steps / dt
        line = {
            "metric": "mll_grad_steps_per_s", "value": round(steps_per_s, 4),
            "unit": f"steps/s (1 step = MLL+grad over {args.batch} series of N={n})",
            "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "ms_per_step_by_rank": [round(t / args.steps * 1e3, 3) for t in dt_ranks],
            "higher_is_better": True, "scaling": args.scaling, "other_scaling": other,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"MLL+grad step, N={n}, batch={B} series per GPU, Volatility kernel + EWMA(k={EWMA_K}) mean",
                       "series_total": per_step_series, "parallelism": f"series-sharded x{world}",
                       "collective": f"all_reduce(2 floats)/step over {backend}" if world > 1 else "none"},
            "step_tflops": round(per_step_series * 2 * n ** 3 / 3 / (dt / args.steps) / 1e12, 2),
            "loss": round(loss, 6), "not_pd": bad,
            "roofline": roof,
            "roofline_step": {"what": "whole step as timed: algorithmic 2N^3/3 flop per series / ms_per_step",
                              "bound": "mfma", "achieved": round(per_step_series * 2 * n ** 3 / 3 / (dt / args.steps) / 1e12, 2),
                              "peak": FP32_MFMA_PEAK_TF * world, "unit": "TFLOP/s",
                              "frac": round(B * 2 * n ** 3 / 3 / (dt / args.steps) / 1e12 / FP32_MFMA_PEAK_TF, 4)},
            "schedule": {"step_kernel": (roof or {}).get("kernel"), "launches_per_step": (roof or {}).get("launches", 0),
                         "streams": "the caller's stream (one launch)" if (roof or {}).get("launches", 0) == 1 else
                                    "library-internal, forked/joined on the caller's stream"},
            "cpu_baseline": cpu,
            "api_step": api,
            "fp64": f64,
            "rollouts": roll,
            "configs": cfgs,
            "next_rows": nxt,
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()                       # rank 0 ran the roofline legs: leave together
        dist.destroy_process_group()


def pmc_traffic(kernel: str, n: int, B: int):
    """HBM bytes per launch of `kernel` from the PMC passes (scripts/pmc.sh + scripts/pmc_traffic.py write
    profiles/pmc_traffic.json with the workload and the hash of the library sources they were taken on).  Returned only
    when workload AND source hash match what is running now -- else (None, why)."""
    from volt_amd import _lib
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return None, "no profiles/pmc_traffic.json"
    if pj.get("config") != {"n": n, "batch": B}:
        return None, f"profiles/pmc_traffic.json is for {pj.get('config')}"
    have = _lib.lib().volt_source_hash().decode()
    if pj.get("source_hash") != have:
        return None, f"profiles/pmc_traffic.json was taken on sources {pj.get('source_hash')}, running {have}"
    k = pj.get("kernels", {}).get(kernel)
    if not k:
        return None, f"profiles/pmc_traffic.json has no kernel {kernel}"
    return round(k["bytes_per_launch"]), f"profiles/pmc_traffic.json ({pj.get('date')}, sources {have}, {k['launches_profiled']} launches)"


def profile_only_leg(N, K, y, raw_noise, ws, n, B, dev):
    """--profile-only N: exactly N gradient steps through volt_profile_step_f32, nothing else on the device."""
    from volt_amd import _lib, ops
    L = _lib.lib()
    s2 = (torch.nn.functional.softplus(raw_noise.detach()) + 1e-4).contiguous()
    inf = torch.empty(B, dtype=torch.int32, device=dev)
    resid_p = (y - ops.ewma(y, EWMA_K)[..., :-1]).contiguous()
    torch.cuda.synchronize()
    ms_sum, ms_un, cnt = (ctypes.c_float * 2)(), (ctypes.c_float * 2)(), (ctypes.c_int * 2)()
    tot = np.zeros(2)
    for _ in range(N):
        _lib.check(L.volt_profile_step_f32(K.data_ptr(), n, n * n, resid_p.data_ptr(), s2.data_ptr(), ws.out.data_ptr(),
                                           ws.alpha.data_ptr(), ws.ptr, inf.data_ptr(), B, n, 0, _lib.stream_ptr(), ms_sum,
                                           ms_un, cnt, None), "profile")
        tot += np.array(list(ms_un))
    Np = ops.padded_n(n)
    fl = kernel_class_flops(B, Np)
    one = cnt[0] == 1 and cnt[1] == 0
    cls_ms = float(tot[0] / N) if one else float(tot.sum() / N)
    print(json.dumps({"profile_only": N, "n": n, "batch": B, "launches_per_step": list(cnt), "class_ms": [round(float(t / N), 4) for t in tot],
                      "step_kernels_ms": round(cls_ms, 4), "achieved_TFLOPs": round(sum(fl) / (cls_ms * 1e-3) / 1e12, 2),
                      "frac": round(sum(fl) / (cls_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4), "not_pd": int((inf != 0).sum().item()),
                      "source_hash": L.volt_source_hash().decode()}), flush=True)


FP64_MFMA_PEAK_TF = 78.6     # v_mfma_f64_16x16x4_f64, dense (vendor figure; 64 cycles per instruction per SIMD)


def f64_leg(x, F, vol, dev, n, B):
    """The fp64 path (a double-precision model keeps its dtype, voltron/kernels/VolKernel.py:28-33): one batched
    factorisation (volt_potrf_f64, N^3/3 flop per matrix) and one MLL+grad step (volt_mll_step_f64, 2N^3/3) on B series."""
    from volt_amd import _lib, ops
    xd = torch.tensor(x, device=dev, dtype=torch.float64)
    K = ops.fill(ops.cumtrapz(torch.tensor(vol[:B], device=dev, dtype=torch.float64), xd, square=True))
    s2 = torch.full((B,), 0.6932, device=dev, dtype=torch.float64)
    r = torch.log(torch.tensor(F[:B, 1:], device=dev, dtype=torch.float64))
    r = r - r.mean(-1, keepdim=True)
    Np = ops.padded_n(n)
    f = ops.potrf(K, s2)
    Aprep = torch.empty_like(f.A)
    L = _lib.lib()
    _lib.check(L.volt_prepare_f64(K.data_ptr(), n, n * n, s2.data_ptr(), 0.0, Aprep.data_ptr(), B, n, _lib.stream_ptr()), "prepare")
    ws = ops.MllWorkspace(B, n, True, dev, torch.float64)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timeit(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def potrf():
        f.A.copy_(Aprep)
        ops.potrf_f64_inplace(f.A, f.Winv, f.info)
    t_copy = timeit(lambda: f.A.copy_(Aprep))
    t_potrf = timeit(potrf) - t_copy
    t_step = timeit(lambda: ops.mll_step(K, r, s2, ws))
    fl = B * Np ** 3 / 3
    # ONE series: the latency chain alone (32 diagonal blocks in a row; csrc/batch64_step.hip)
    f1 = ops.potrf(K[:1], s2[:1])
    A1 = f1.A.clone()

    def potrf1():
        f1.A.copy_(A1prep)
        ops.potrf_f64_inplace(f1.A, f1.Winv, f1.info)
    A1prep = Aprep[:1].clone()
    t_copy1 = timeit(lambda: f1.A.copy_(A1prep), reps=10)
    t_potrf1 = timeit(potrf1, reps=10) - t_copy1
    one_launch = bool(L.volt_potrf_workspace_bytes_f64(B, Np))
    return {"workload": f"{B} series of N={n} in fp64 (v_mfma_f64_16x16x4_f64)",
            "schedule": "one launch (csrc/batch64_step.hip)" if one_launch else "launch per block column (csrc/chol64.hip)",
            "single_series_potrf_ms": round(t_potrf1, 3),
            "potrf_ms": round(t_potrf, 3), "potrf_tflops": round(fl / t_potrf / 1e9, 2),
            "potrf_frac_of_fp64_mfma_peak": round(fl / t_potrf / 1e9 / FP64_MFMA_PEAK_TF, 4),
            "mll_grad_step_ms": round(t_step, 3), "mll_grad_step_tflops": round(2 * fl / t_step / 1e9, 2),
            "mll_grad_step_frac_of_fp64_mfma_peak": round(2 * fl / t_step / 1e9 / FP64_MFMA_PEAK_TF, 4),
            "not_pd": int((ws.info != 0).sum().item())}


def api_leg(x, F, vol, dev, n, B, raw_ms, t1=6, t2=26):
    """The drop-in surface at the metric size: one iteration of TrainVoltMagpieBatch -- ``loss = -mll(model(x), y);
    loss.backward(); optimizer.step()``, voltron/train_utils.py:243-254 -- for B series of length n, per iteration as
    the difference of two runs (t2 vs t1 iterations: construction and the one-off fill cancel).  eager = the loop with
    the factorisation's info check deferred (the default of the batched loop); eager_per_step_check = with the host
    read-back every step, as a literal gpytorch loop does; graph = graph=True (one captured iteration replayed where the
    step is launch-bound).  default = no `graph` argument: what a caller of the reference's signature gets."""
    from volt_amd.train_utils import TrainVoltMagpieBatch
    tx = torch.tensor(x, device=dev)
    prices = torch.tensor(F[:B, 1:], device=dev)
    v = torch.tensor(vol[:B], device=dev)

    def run(iters, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        TrainVoltMagpieBatch(tx, prices, v, train_iters=iters, k=EWMA_K, **kw)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    out = {"workload": f"TrainVoltMagpieBatch iteration, {B} x N={n}, EWMA(k={EWMA_K}) mean, Adam on raw_noise [B]",
           "raw_op_ms_per_step": round(raw_ms, 3), "iterations": [t1, t2]}
    from volt_amd.train_utils import _capture_pays
    # default = what a drop-in caller gets (no reference call site passes `graph`: captured where the step is launch-bound)
    for mode, kw in (("default", {}), ("eager", {"graph": False}), ("eager_per_step_check", {"graph": False, "defer": False}),
                     ("graph", {"graph": True})):
        run(t1, **kw)                                  # warm: allocator, schedule tables, graph pools
        a = min(run(t1, **kw) for _ in range(2))
        b = min(run(t2, **kw) for _ in range(2))
        ms = (b - a) / (t2 - t1) * 1e3
        out[mode] = {"ms_per_step": round(ms, 3), "overhead_vs_raw_op": round(ms / raw_ms - 1.0, 4)}
    out["default"]["captured"] = bool(_capture_pays(prices))
    out["ref_default_1x399_api"] = api_default_leg(dev)
    return out


def api_default_leg(dev, n=399, t1=30, t2=230):
    """The reference's own default size through its own entry point with its own arguments: TrainVoltMagpieModel(train_x,
    train_y, vol_model, vol_lh, vol_path, train_iters) for ONE series of ntrain = 400 prices (experiments/stocks/
    ForecastGenerator.py:53-91) -- no `graph` argument, as at every reference call site (voltron/train_utils.py:192).  Per
    iteration as the difference of two runs; beside it the raw op (ops.mll_step alone, K resident) in a tight loop."""
    from volt_amd import ops
    from volt_amd.synthetic import sde_batch
    from volt_amd.train_utils import TrainVoltMagpieModel, _capture_pays
    x, F, vol = sde_batch(1, n, seed=2019)
    tx = torch.tensor(x, device=dev)
    prices = torch.tensor(F[0], device=dev)
    v = torch.tensor(vol[0], device=dev)

    def run(iters, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        TrainVoltMagpieModel(tx, prices[1:], None, None, v, train_iters=iters, k=EWMA_K, **kw)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    res = {}
    for mode, kw in (("default", {}), ("eager", {"graph": False})):
        run(t1, **kw)
        a = min(run(t1, **kw) for _ in range(3))
        b = min(run(t2, **kw) for _ in range(3))
        res[mode] = (b - a) / (t2 - t1) * 1e3
    K = ops.fill(ops.cumtrapz(v[None], tx, square=True))
    y = torch.log(prices[1:])[None]
    r = (y - y.mean(-1, keepdim=True)).contiguous()
    s2 = torch.full((1,), 0.6932, device=dev)
    ws = ops.MllWorkspace(1, n, True, dev)
    for _ in range(5):
        ops.mll_step(K, r, s2, ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.mll_step(K, r, s2, ws)
    e1.record()
    torch.cuda.synchronize()
    raw = e0.elapsed_time(e1) / 200
    return {"workload": f"TrainVoltMagpieModel, 1 x N={n}, reference arguments (no `graph`)", "raw_op_ms_per_step": round(raw, 4),
            "ms_per_iteration": round(res["default"], 4), "over_raw_op": round(res["default"] / raw, 3),
            "captured_by_default": bool(_capture_pays(y)), "eager_ms_per_iteration": round(res["eager"], 4),
            "iterations": [t1, t2]}


def next_rows_leg(dev):
    """SURVEY 8(f) rows 1 and 4 on the record: the step of the two fits that precede the data model in the
    reference's per-window pipeline (experiments/stocks/GenerateMultiMeanPreds.py:100-111) --
      * LearnGPCV's ELBO + gradient step (voltron/train_utils.py:15-67): volt_gpcv_step_f32, algorithmic 5 N^3 / 3 flop per
        series (factor + L^-T through the exact-GP step's one-launch schedules, T' = Lq' L^-T, G = K^-1 Lq);
      * TrainVolModel's MLL + gradient step (:69-95): volt_mll_step_f32 on K = vol * min(x, x'), 2 N^3 / 3 (the gradient wrt
        the kernel's scale is in closed form from the step's scalars: no dK contraction) --
    at the reference's default size for one ticker and for 64, and at 8 x 4096: ms of the raw HIP step (K resident, tight
    loop, HIP events), and ms per iteration of the public entry point with the reference's arguments (captured where the
    step is launch-bound; difference of two runs).  frac = algorithmic flops / time / the fp32 MFMA peak."""
    import math
    from volt_amd import ops
    from volt_amd.synthetic import sde_batch
    from volt_amd.train_utils import LearnGPCV, TrainVolModel, TrainVolModelBatch
    from volt_amd.variational import _gauss_hermite
    out = {}
    for B, n in ((1, 399), (64, 399), (8, 4096)):
        x, F, vol = sde_batch(B, n, seed=5)
        tx = torch.tensor(x, device=dev)
        prices = torch.tensor(F, device=dev)                                  # [B, n + 1]
        volp = torch.tensor(vol, device=dev).clamp_min(1e-3)
        dtx = float(x[1] - x[0])
        yy = (prices[:, 1:] - prices[:, :-1]) / prices[:, :-1] / dtx ** 0.5   # scaled returns (train_utils.py:16-18)
        K = 0.2 * torch.minimum(tx[:, None], tx[None, :]).expand(B, n, n).contiguous() + 0.0
        Lq = (0.05 * torch.eye(n, device=dev) + 0.001 * torch.randn(n, n, device=dev, generator=torch.Generator(dev).manual_seed(0))).tril()
        Lq = Lq.expand(B, n, n).contiguous()
        m = yy.abs().clamp_min(1e-2).log()
        mu = torch.full((B, n), -1.5, device=dev)
        gx, gw = _gauss_hermite(75, dev)
        reps = 100 if n < 1000 else 10

        def timed_loop(fn):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        gws = ops.gpcv_step(K, m - mu, m, Lq, yy, gx, gw, w_ell=1 / n, w_kl=1 / n)
        g_ms = timed_loop(lambda: ops.gpcv_step(K, m - mu, m, Lq, yy, gx, gw, gws, w_ell=1 / n, w_kl=1 / n))
        g_bad = int((gws.info != 0).sum().item())
        del gws
        lv = volp.log()
        r = (lv - lv.mean(-1, keepdim=True)).contiguous()
        s2 = torch.full((B,), 0.6932, device=dev)
        mws = ops.MllWorkspace(B, n, True, dev)
        v_ms = timed_loop(lambda: ops.mll_step(K, r, s2, mws))
        v_bad = int((mws.info != 0).sum().item())
        del mws

        def api(fn, t1, t2):
            def run(iters):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn(iters)
                torch.cuda.synchronize()
                return time.perf_counter() - t0
            run(t1)
            a = min(run(t1) for _ in range(2))
            b = min(run(t2) for _ in range(2))
            return (b - a) / (t2 - t1) * 1e3
        t1, t2 = (20, 120) if n < 1000 else (5, 13)
        py = prices[0] if B == 1 else prices
        vp = volp[0] if B == 1 else volp
        try:
            ga = api(lambda it: LearnGPCV(tx, py, train_iters=it), t1, t2)
            va = api(lambda it: (TrainVolModel if B == 1 else TrainVolModelBatch)(tx, vp, train_iters=it), t1, t2)
        except Exception as e:                                                # never let the extra leg cost the line
            ga = va = None
            out.setdefault("errors", []).append(repr(e)[:200])
        gfl, vfl = B * 5 * n ** 3 / 3, B * 2 * n ** 3 / 3
        row = lambda ms, fl: None if ms is None else {"ms": round(ms, 4), "tflops": round(fl / (ms * 1e-3) / 1e12, 2),
                                                      "frac": round(fl / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4)}
        out[f"{B}x{n}"] = {"gpcv_elbo_step": {"algorithmic_flops": "5 N^3 / 3 per series", "raw_step": row(g_ms, gfl),
                                              "LearnGPCV_iteration": row(ga, gfl), "not_pd": g_bad},
                           "vol_model_step": {"algorithmic_flops": "2 N^3 / 3 per series", "raw_step": row(v_ms, vfl),
                                              "TrainVolModel_iteration": row(va, vfl), "not_pd": v_bad}}
        del K, Lq
    out["what"] = ("SURVEY 8(f) rows 1 (TrainVolModel, voltron/train_utils.py:69-95) and 4 (LearnGPCV, :15-67): raw HIP step and one "
                   "iteration of the public entry point with the reference's arguments")
    return out


def configs_leg(make_step, timed, dev, K_all, y_all, n_head, batch_head, steps=50, warmup=5, api_inputs=None):
    """Every BASELINE.json configuration beside the metric (and the reference's own default size), driver-run: the
    training-loop body of the headline (EWMA mean -> softplus -> volt_mll_step_f32 -> chain rule -> Adam) through the same
    timing loop, `steps` steps each.  frac = algorithmic 2 N^3 / 3 flop per series / ms_per_step / the fp32 MFMA peak."""
    from volt_amd import ops
    from volt_amd.synthetic import sde_batch
    shapes = [("C2_1x4096", 1, 4096, "BASELINE config 2: one series, N=4096 (voltron/train_utils.py:243-254 trains exactly one)"),
              ("C3_64x2048", 64, 2048, "BASELINE config 3: 64 tickers, N=2048"),
              ("C4_share_32x4096", 32, 4096, "BASELINE config 4 (256 stations over 8 GPUs): the per-GPU share"),
              ("B8_8x4096", 8, 4096, "the metric (64 x 4096) strong-scaled over 8 GPUs: the per-GPU share"),
              ("B16_16x4096", 16, 4096, "the metric strong-scaled over 4 GPUs: the per-GPU share"),
              ("ref_default_64x399", 64, 399, "the reference's default ntrain=400 (experiments/stocks/ForecastGenerator.py), 64 tickers")]
    out = {}
    for name, Bc, nc, what in shapes:
        if nc == n_head and Bc <= K_all.shape[0]:
            Ks, ys = K_all[:Bc], y_all[:Bc]
        else:
            x, F, vol = sde_batch(Bc, nc, seed=2019)
            Ks = ops.fill(ops.cumtrapz(torch.tensor(vol, device=dev), torch.tensor(x, device=dev), square=True))
            ys = torch.log(torch.tensor(F[:, 1:], device=dev))
        step, _, ws_, _ = make_step(Bc, Ks, ys, nc)
        dt, _, info = timed(step, warmup, steps)
        ms = dt / steps * 1e3
        tf = Bc * 2 * nc ** 3 / 3 / (ms * 1e-3) / 1e12
        out[name] = {"what": what, "batch": Bc, "n": nc, "steps": steps, "ms_per_step": round(ms, 4), "tflops": round(tf, 2),
                     "frac": round(tf / FP32_MFMA_PEAK_TF, 4), "not_pd": int((info != 0).sum().item())}
        del step, ws_, Ks, ys
        # the same shape through the drop-in surface with the reference's arguments (no `graph`: captured where the step is
        # launch-bound, volt_amd/train_utils.py) -- per iteration as the difference of two runs
        try:
            from volt_amd.train_utils import TrainVoltMagpieBatch, _capture_pays
            if nc == n_head and Bc <= K_all.shape[0]:
                xa, Fa, va = api_inputs
                xa, Fa, va = xa, Fa[:Bc], va[:Bc]
            else:
                xa, Fa, va = x, F, vol
            tx = torch.tensor(xa, device=dev)
            pr = torch.tensor(Fa[:Bc, 1:], device=dev)
            vv = torch.tensor(va[:Bc], device=dev)

            def run(iters):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                TrainVoltMagpieBatch(tx, pr, vv, train_iters=iters, k=EWMA_K)
                torch.cuda.synchronize()
                return time.perf_counter() - t0
            t1_, t2_ = (10, 60) if ms < 2.0 else (6, 16)
            run(t1_)
            a_ = min(run(t1_) for _ in range(2))
            b_ = min(run(t2_) for _ in range(2))
            api_ms = (b_ - a_) / (t2_ - t1_) * 1e3
            out[name]["api_default"] = {"ms_per_iteration": round(api_ms, 4), "captured": bool(_capture_pays(pr)),
                                        "tflops": round(Bc * 2 * nc ** 3 / 3 / (api_ms * 1e-3) / 1e12, 2),
                                        "frac": round(Bc * 2 * nc ** 3 / 3 / (api_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4)}
        except Exception as e:                          # never let the extra leg cost the line
            out[name]["api_default"] = {"error": repr(e)[:200]}
    return out


# VALU issue model of the rollout kernel (round 4): the kernel is neither HBM- nor MFMA-bound -- one wave walks H
# dependent steps of a path -- so its roofline is the rate at which the chip issues wave-instructions:
# 256 CUs x 4 SIMDs x 1 VALU wave-instruction per 2 cycles (MI355X_MICROARCH.md, per-instruction cycle constants:
# v_fma_f32 wave64 = 2 cycles throughput) at 2.4 GHz = 1.23e12 / s.  `achieved` counts the kernel's VALU wave-instructions from the PMC pass kept in profiles/r04/ (SQ_INSTS_VALU).
VALU_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 2


def rollout_leg(x, F, vol, dev, n, G=8, S=10000, H=256, dist=None, world=1, rank=0):
    """BASELINE config 5 (10k sample paths x 256-step horizon, 64 series over 8 GPUs).
    The engine as shipped: per-sample bordered factor extended by ONE entry per step (the volatility kernel's
    cross-covariance prefix is step-invariant).  The train block enters through two scalars per series; `total_s` takes
    them the way the REFERENCE does (rollout_utils.py:35-36: factor the noise-free train block -- here in fp64 on
    volt_potrf_f64 -- and two solves: that work is inside the total); `closed_form` is the same run with K^-1 u = e_last
    taken analytically (an identity of this kernel, SURVEY 4: reserved for testing, reported beside).  Algorithmic HBM
    bytes of the kernel: pred_vol and z read, samples written = 12 B per sample-step.  `resubstitute` = the same paths
    with every sample's triangular system re-solved from its stored rows at every step, the round-2 engine: H^3/6 * 4 B
    per path that an append-only w_s does not need -- reported as `redundant_bytes`, credited to nothing.
    world > 1: every rank runs its G series; rank 0 reports the job's aggregate and the timed gather."""
    from volt_amd import rollout_engine as re_
    from volt_amd import distributed as vd
    from volt_amd.synthetic import rollout_inputs
    G = min(G, F.shape[0])
    pv, z = rollout_inputs(vol[:G, -1], S, H, seed=3 + rank)
    tx = torch.tensor(x, device=dev)
    test_x = torch.arange(H, device=dev) / 252. + tx[-1] + tx[1]
    logy = torch.log(torch.tensor(F[:G, 1:], device=dev))
    lv = torch.log(torch.tensor(vol[:G], device=dev))
    pvd, zd = torch.tensor(pv, device=dev), torch.tensor(z, device=dev)

    def run(**kw):
        best_total, best_kernel, info = None, None, None
        for rep in range(3):
            tm = {}
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            samples, info = re_.rollout_series(tx, logy, lv, test_x, pvd, zd, 0, EWMA_K, timing=tm, **kw)
            torch.cuda.synchronize()
            tot = time.perf_counter() - t0
            ker = tm["start"].elapsed_time(tm["stop"]) * 1e-3
            if rep and (best_total is None or tot < best_total):
                best_total, best_kernel = tot, ker
        bad = int((info != 0).sum().item())
        return best_total, best_kernel, bad, samples

    tot_c, ker_c, bad_c, smp_c = run(solve="closed")
    tot_f, ker_f, bad_f, smp_f = run(solve="factor")
    dev_cf = float((smp_c - smp_f).abs().max().item())
    multi = None
    if dist is not None:
        # the job: every rank's series / the slowest rank; then the gather of the samples every rank holds
        tt = torch.tensor([tot_f, tot_c, float(bad_f)], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        gathers = []
        for rep in range(3):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            parts = vd.gather_samples(smp_f)
            torch.cuda.synchronize()
            gathers.append(time.perf_counter() - t0)
        gt = torch.tensor([min(gathers[1:])], device=dev, dtype=torch.float64)
        allg = [torch.zeros_like(gt) for _ in range(world)]
        dist.all_gather(allg, gt)
        nbytes = sum(int(p_.numel()) * 4 for p_ in parts)
        slow_f = max(float(t[0]) for t in allt)
        slow_c = max(float(t[1]) for t in allt)
        g_s = max(float(g[0]) for g in allg)
        multi = {"ranks": world, "series_per_rank": G, "series_total": G * world,
                 "total_s_by_rank": [round(float(t[0]), 5) for t in allt], "total_s": round(slow_f, 5),
                 "aggregate_sample_steps_per_s": round(world * G * S * H / slow_f),
                 "closed_form_total_s": round(slow_c, 5), "non_pd_paths": int(sum(float(t[2]) for t in allt)),
                 "gather_samples_s": round(g_s, 5), "gathered_bytes_per_rank": nbytes,
                 "gather_GBps_per_rank": round(nbytes * (world - 1) / world / g_s / 1e9, 2),
                 "with_gather_sample_steps_per_s": round(world * G * S * H / (slow_f + g_s)),
                 "note": "every rank rolls out its own series (no collective in the rollout); gather_samples = "
                         "volt_amd.distributed.gather_samples (all_gather of [G,S,H] fp32), max over ranks, best of 2"}
        del parts
    del smp_f
    if rank != 0:
        return None
    res = None
    if dist is None:
        tot_r, ker_r, bad_r, smp_r = run(solve="closed", resubstitute=True)
        same = float((smp_c - smp_r).abs().max().item())    # (rounds 3 - 5: bitwise; round 6's engine sums w_s'w_s in fp64: to rounding)
        del smp_r
        red = G * S * H ** 3 / 6 * 4
        res = {"total_s": round(tot_r, 5), "kernel_ms": round(ker_r * 1e3, 3), "non_pd_paths": bad_r,
               "max_abs_dev_from_default": same, "redundant_bytes": int(red),
               "streamed_GBps": round(red / ker_r / 1e9, 1),
               "frac_of_hbm_peak_on_redundant_bytes": round(red / ker_r / 1e9 / HBM_PEAK_GBS, 4)}
    del smp_c
    alg = G * S * H * 12.0
    # VALU wave-instructions per sample-step of the rollout kernel, from the PMC pass scripts/pmc_rollout_issue.sh writes to
    # profiles/rollout_pmc.json (SQ_INSTS_VALU / (G S H)) together with the shape AND the hash of the library sources the counter
    # was taken on: emitted only when both match what is running (a counter of another binary is not evidence: VERDICT r5 weak 4)
    issue, issue_why = None, None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "rollout_pmc.json")))
        # the per-sample-step count was measured at ONE shape (it grows with H: a step's work is O(steps so far))
        if pj.get("shape") != {"G": G, "S": S, "H": H, "n": n}:
            raise ValueError("profiles/rollout_pmc.json is for another shape")
        from volt_amd import _lib
        have = _lib.lib().volt_source_hash().decode()
        if pj.get("source_hash") != have:
            raise ValueError(f"profiles/rollout_pmc.json was taken on sources {pj.get('source_hash')}, running {have}")
        per = float(pj["valu_wave_insts_per_sample_step"])
        ach = G * S * H * per / ker_f
        issue = {"kernel": pj.get("kernel", "rollout_bordered_kernel<1, false, 0>"), "bound": "valu_issue", "achieved": round(ach / 1e12, 3),
                 "peak": round(VALU_WAVE_INSTS_PER_S / 1e12, 3), "unit": "T wave-instructions/s",
                 "frac": round(ach / VALU_WAVE_INSTS_PER_S, 4), "valu_wave_insts_per_sample_step": per,
                 "source": f"instruction count per sample-step: profiles/rollout_pmc.json ({pj.get('date')}, rocprofv3 --pmc SQ_INSTS_VALU "
                           f"at this very shape on sources {have} = the running library); kernel time: live (HIP events)"}
    except Exception as err:
        issue, issue_why = None, str(err)
    return {"workload": f"{G} series x {S} paths x {H} steps, N={n} (BASELINE config 5"
                        + (", per-GPU share" if dist is None else f", {G} of 64 series on each of {world} ranks")
                        + "); append-only bordered engine; train block by the reference's route (fp64 factorisation + two "
                          "solves, inside total_s) with the closed form beside it",
            "solve": "factor", "total_s": round(tot_f, 5), "kernel_ms": round(ker_f * 1e3, 3),
            "sample_steps_per_s": round(G * S * H / tot_f), "non_pd_paths": bad_f,
            "includes": f"volt_potrf_f64 of {G} x {n}^2 + two fp64 triangular solves per series",
            "closed_form": {"total_s": round(tot_c, 5), "kernel_ms": round(ker_c * 1e3, 3),
                            "sample_steps_per_s": round(G * S * H / tot_c), "non_pd_paths": bad_c,
                            "max_abs_dev_from_factor_route": dev_cf},
            "roofline": issue, "roofline_absent_because": issue_why,
            "hbm": {"algorithmic_GB": round(alg / 1e9, 3), "GBps": round(alg / ker_f / 1e9, 1),
                    "note": "12 B per sample-step: far from HBM-bound (one wave per path, H dependent steps each)"},
            "resubstitute": res, "multi_rank": multi}


def _cpu_worker(idx, threads, Kc, yc, mc, reps, barrier, q):
    """One worker of the whole-host CPU baseline: its own series, `threads` intra-op threads, all workers in step."""
    import torch as th
    from oracle import torch_cpu_path as tp
    th.set_num_threads(threads)
    raw = th.full((Kc.shape[0],), 1e-5, requires_grad=True)
    tp.mll_step(Kc, yc, mc, raw)                      # warm-up (also sizes the allocator)
    barrier.wait()
    t0 = time.perf_counter()
    for _ in range(reps):
        tp.mll_step(Kc, yc, mc, raw)
    t1 = time.perf_counter()
    q.put((idx, t0, t1))


def cpu_baseline_leg(K, y, mean, n, batch):
    """The CPU path timed on this box's host cores on a bounded sample of the same workload: the torch-only
    restatement of gpytorch's dense step (kind "port"), and real gpytorch under max_cholesky_size(N+1) when it is
    importable -- reported, never the optimisation target.

    Two figures.  `single_worker`: one process, the series in one batched call, at the thread count that is fastest for
    ONE series (LAPACK/MKL does not scale to every core of a big host at this size).  `value` (the headline baseline):
    the WHOLE host -- cpu_count // best_threads concurrent worker processes of best_threads threads each over disjoint
    series (the series are independent), aggregate series-steps/s / batch."""
    from oracle import torch_cpu_path as tp
    ncpu = os.cpu_count() or 1
    bs = 8 if n >= 4096 else 16
    bs = min(bs, K.shape[0])
    Kc, yc, mc = K[:bs].cpu(), y[:bs].cpu(), mean[:bs].cpu()
    best_t, best_c = None, None
    for t_ in sorted({c for c in (4, 8, 16, 32, 64, ncpu) if c <= ncpu}):
        torch.set_num_threads(t_)
        r1 = torch.full((1,), 1e-5, requires_grad=True)
        tp.mll_step(Kc[:1], yc[:1], mc[:1], r1)
        t1 = time.perf_counter()
        tp.mll_step(Kc[:1], yc[:1], mc[:1], r1)
        c_ = time.perf_counter() - t1
        if best_c is None or c_ < best_c:
            best_t, best_c = t_, c_
    cores = best_t
    torch.set_num_threads(cores)
    rawc = torch.full((bs,), 1e-5, requires_grad=True)
    tp.mll_step(Kc, yc, mc, rawc)                    # warm-up
    reps = 3
    times = []
    for _ in range(reps):
        t1 = time.perf_counter()
        tp.mll_step(Kc, yc, mc, rawc)
        times.append(time.perf_counter() - t1)
    tc = float(np.median(times))
    single = {"value": round(1.0 / (tc / bs * batch), 5), "unit": "steps/s", "cores": cores,
              "sample": f"{bs} of {batch} series x {reps} steps at N={n} in one batched call (median {tc:.2f} s per "
                        f"{bs}-series step, min {min(times):.2f} max {max(times):.2f}), scaled x{batch / bs:g}; threads "
                        f"calibrated over 4..{ncpu} on one series, best = {cores}"}
    # ---- the whole host: workers x threads = every core, disjoint series, all in flight at once
    import torch.multiprocessing as mp
    nw = max(1, min(ncpu // cores, 64))
    per = 1                                           # series per worker (bounded sample: nw series in flight)
    wreps = 2
    Kh = K[: min(K.shape[0], nw * per)].cpu()
    nw = min(nw, Kh.shape[0] // per)
    yh, mh = y[: nw * per].cpu(), mean[: nw * per].cpu()
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(nw), ctx.Queue()
    procs = []
    for w_ in range(nw):
        sl = slice(w_ * per, (w_ + 1) * per)
        pr = ctx.Process(target=_cpu_worker, args=(w_, cores, Kh[sl].clone().share_memory_(), yh[sl].clone().share_memory_(),
                                                   mh[sl].clone().share_memory_(), wreps, barrier, q))
        pr.start()
        procs.append(pr)
    spans = []
    try:
        for _ in range(nw):
            spans.append(q.get(timeout=900))
    finally:
        for pr in procs:
            pr.join(30)
            if pr.is_alive():
                pr.kill()
    wall = max(s[2] for s in spans) - min(s[1] for s in spans)
    series_steps_per_s = nw * per * wreps / wall
    cpu = {"value": round(series_steps_per_s / batch, 5), "unit": "steps/s", "cores": nw * cores, "kind": "port",
           "sample": f"{nw} concurrent workers x {cores} threads = {nw * cores} of {ncpu} host cores, {per} series of N={n} "
                     f"each x {wreps} steps (torch-CPU fp32 cholesky+autograd), wall {wall:.2f} s for {nw * per * wreps} "
                     f"series-steps, scaled to the batch of {batch}",
           "single_worker": single}
    torch.set_num_threads(cores)
    cpu["gpytorch"] = gpytorch_leg(Kc, yc, mc, n, batch, cores)
    if cpu["gpytorch"].get("value") is None:
        cpu["note"] = "gpytorch absent -- baseline is a torch-only restatement (oracle/torch_cpu_path.py)"
    return cpu


def gpytorch_leg(Kc, yc, mc, n, batch, cores):
    """Real gpytorch, if this box has it (it is not in the build image): the reference's own step
    ``loss = -mll(model(x), y); loss.backward()`` (voltron/train_utils.py:243-250) with the cached train_cov as the
    model's covariance, one series at a time as the reference trains, Cholesky path forced by
    max_cholesky_size(N+1).  Returns {"value": None, "reason": ...} when it cannot run."""
    try:
        import gpytorch
    except Exception as e:                                  # noqa: BLE001 -- absent or broken: report, do not fail
        return {"value": None, "reason": f"import gpytorch failed: {type(e).__name__}: {e}"[:200]}
    try:
        class _Cached(gpytorch.models.ExactGP):
            def __init__(self, tx, ty, lik, mean, cov):
                super().__init__(tx, ty, lik)
                self._m, self._c = mean, cov

            def forward(self, x):
                return gpytorch.distributions.MultivariateNormal(self._m, self._c)

        torch.set_num_threads(cores)
        ns = min(2, Kc.shape[0])
        tx = torch.arange(n, dtype=torch.float32) / 252.
        times, vals = [], []
        with gpytorch.settings.max_cholesky_size(n + 1):
            for b in range(ns):
                lik = gpytorch.likelihoods.GaussianLikelihood()
                lik.raw_noise.data = torch.tensor([1e-5])                       # train_utils.py:222
                model = _Cached(tx, yc[b], lik, mc[b], Kc[b])
                model.train(); lik.train()
                mll = gpytorch.mlls.ExactMarginalLogLikelihood(lik, model)
                for rep in range(2):
                    lik.zero_grad()
                    t1 = time.perf_counter()
                    loss = -mll(model(tx), yc[b])
                    loss.backward()
                    if rep:
                        times.append(time.perf_counter() - t1)
                vals.append(float(-loss))
        tc = float(np.median(times))
        return {"value": round(1.0 / (tc * batch), 5), "unit": "steps/s", "cores": cores, "kind": "reference",
                "version": getattr(gpytorch, "__version__", "?"), "mll_first_series": vals[0],
                "sample": f"{ns} of {batch} series x 1 step at N={n} (gpytorch ExactMarginalLogLikelihood + backward under "
                          f"max_cholesky_size(N+1), {tc:.2f} s per series), scaled x{batch} to the batch"}
    except Exception as e:                                  # noqa: BLE001
        return {"value": None, "reason": f"gpytorch present but the leg failed: {type(e).__name__}: {e}"[:300]}


if __name__ == "__main__":
    main()
