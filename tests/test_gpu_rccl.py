"""Multi-GPU readiness: `python bench.py --gpus N` starts its own ranks and refuses to print an N-GPU line
from fewer devices; the RCCL ("nccl" backend) path is exercised as soon as a box has two GPUs.

On the 1-GPU test box: the refusal, and the self-launch through the dry-run hooks (both ranks on device 0 over gloo).
On a >= 2-GPU box: two ranks on two devices over RCCL, the reduced loss / gradient against a single-process run."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "3", "--warmup", "1", "--series-len", "512", "--batch", "16", "--no-cpu-baseline", "--no-aux-legs",
         "--no-rollouts"]


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra)
    return env


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with fewer GPUs than asked for")
def test_bench_refuses_more_gpus_than_present():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, env=_clean_env(), cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "needs 2 visible GPUs" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]      # no N-GPU line from one rank


def test_bench_launches_its_own_ranks():
    """No torchrun around it: bench.py --gpus 2 re-execs under torch.distributed.run (dry-run: one device, gloo)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL,
                         env=_clean_env(VOLT_BENCH_ONE_DEVICE="1", VOLT_BENCH_BACKEND="gloo"), cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and len(d["ms_per_step_by_rank"]) == 2
    assert d["scaling"] == "weak" and d["config"]["series_total"] == 32
    o = d["other_scaling"]
    assert o["scaling"] == "strong" and o["series_total"] == 16 and o["not_pd"] == 0
    assert "gloo" in d["config"]["collective"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rccl_worker(rank, world, port, total, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from volt_amd import distributed as vd, ops
        from volt_amd.synthetic import sde_batch
        lo, hi = vd.shard_range(total)
        x, F, vol = sde_batch(hi - lo, n, seed=2019, first=lo)
        K = ops.fill(ops.cumtrapz(torch.tensor(vol, device=dev), torch.tensor(x, device=dev), square=True))
        y = torch.log(torch.tensor(F[:, 1:], device=dev))
        resid = y - ops.ewma(y, 25)[..., :-1]
        s2 = torch.full((hi - lo,), 0.6932, device=dev)
        out, _, info = ops.mll_step(K, resid, s2)
        local = torch.stack([out[:, 0].sum(), out[:, 1].sum(), torch.tensor(float(hi - lo), device=dev)])
        red = vd.all_reduce_scalars(local)
        q.put((rank, red.cpu().tolist(), int(info.abs().sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_rccl_two_ranks_reduced_loss_equals_single_process_sum():
    import torch.multiprocessing as mp
    from volt_amd import ops
    from volt_amd.synthetic import sde_batch
    world, total, n = 2, 6, 640
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, total, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2] == 0
    dev = torch.device("cuda", 0)
    x, F, vol = sde_batch(total, n, seed=2019)
    K = ops.fill(ops.cumtrapz(torch.tensor(vol, device=dev), torch.tensor(x, device=dev), square=True))
    y = torch.log(torch.tensor(F[:, 1:], device=dev))
    out, _, _ = ops.mll_step(K, y - ops.ewma(y, 25)[..., :-1], torch.full((total,), 0.6932, device=dev))
    np.testing.assert_allclose(res[0][1][0], float(out[:, 0].sum()), rtol=1e-5)
    np.testing.assert_allclose(res[0][1][1], float(out[:, 1].sum()), rtol=1e-4)
    assert res[0][1][2] == total


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_bench_two_gpus_over_rccl():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, env=_clean_env(), cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and "nccl" in d["config"]["collective"]
