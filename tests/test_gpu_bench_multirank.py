"""bench.py's N > 1 path on a 1-GPU box: two ranks launched the way the driver launches them (torch.distributed.run,
127.0.0.1 rendezvous), both on device 0 over gloo (VOLT_BENCH_ONE_DEVICE / VOLT_BENCH_BACKEND dry-run hooks), in both
scaling modes.  Checks the contract's JSON line: one line from rank 0, whole-job value, series split as stated."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks(scaling):
    env = dict(os.environ, VOLT_BENCH_ONE_DEVICE="1", VOLT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--series-len", "512", "--batch", "16", "--scaling", scaling, "--no-cpu-baseline", "--no-aux-legs", "--with-rollouts",
           "--rollout-samples", "64", "--rollout-horizon", "16"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                         # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["steps"] == 3 and d["not_pd"] == 0
    assert d["ranks_seen"] == 2 and len(d["ms_per_step_by_rank"]) == 2
    assert d["other_scaling"]["scaling"] == ("strong" if scaling == "weak" else "weak")
    per_gpu = 16 if scaling == "weak" else 8
    assert d["config"]["series_total"] == 2 * per_gpu
    # value counts batches of 16 series per second over the whole job
    assert abs(d["value"] - d["config"]["series_total"] / 16 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-2 * d["value"]
    assert d["config"]["collective"].startswith("all_reduce")
    # BASELINE config 5 across ranks: every rank rolls out its share of the series, the gather is timed on its own
    m = d["rollouts"]["multi_rank"]
    assert m["ranks"] == 2 and m["series_total"] == 2 * m["series_per_rank"] and len(m["total_s_by_rank"]) == 2
    assert m["non_pd_paths"] == 0 and m["gather_samples_s"] > 0
    assert m["gathered_bytes_per_rank"] == m["series_total"] * 64 * 16 * 4
    assert abs(m["aggregate_sample_steps_per_s"] - m["series_total"] * 64 * 16 / m["total_s"]) <= 5e-2 * m["aggregate_sample_steps_per_s"]   # total_s is rounded to 10 us
    assert d["rollouts"]["solve"] == "factor" and "closed_form" in d["rollouts"]
