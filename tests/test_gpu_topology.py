"""Off the full chip: the library's gates were measured on 256 CUs in 8 XCDs; on any other device it
scales the slot gates and keeps to the launch-per-column schedules.  Two subprocesses (the knobs and the device are read
once per process) run 2 x 4096, 8 x 399 and 1 x 1500 -- shapes whose default schedules are the split-K launches, the
short-series one-launch step and the long-series one-launch step -- against the fp64 oracle:
  * planned for a faked 64-CU / 2-XCD device (VOLT_TUNE=1 VOLT_FAKE_CUS=64 VOLT_FAKE_XCCS=2) on the real one;
  * under HSA_CU_MASK (a quarter of the CUs), where the hand-offs of whatever schedule runs must still complete.
Round 6: the one-launch batched steps no longer assume where a workgroup runs -- pullers read their XCC id
and take tickets from the queues that XCD owns (csrc/common.h).  So the same subprocess also runs shapes that take the
fence-free LOCAL hand-offs (8 x 2048 and 64 x 1024 in fp32, 8 x 1024 in fp64: batches that are multiples of 8)
  * under the CU mask (XCDs with no CU at all leave orphan queues that the others adopt),
  * with the queues shifted against the XCD numbers (VOLT_BATCH_XSKEW),
  * with the pullers of some XCDs leaving at once (VOLT_BATCH_XDROP: their queues are adopted whole)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import ctypes as C, json, sys
import numpy as np, torch
sys.path.insert(0, "tests")
from test_gpu_contract import SIG2, _check_vs_oracle, _series_problem, dev
from volt_amd import _lib, ops
L = _lib.lib()
o = (C.c_int * 7)(); L.volt_topology_describe(o)
res = {"topology": list(o), "cus_seen": torch.cuda.get_device_properties(0).multi_processor_count}
shapes = ((2, 4096), (8, 399), (1, 1500), (8, 2048), (64, 1024))
if len(sys.argv) > 1 and sys.argv[1] == "local":
    shapes = ((8, 2048), (64, 1024), (16, 1536))
for B, n in shapes:
    x, vol, y, mean = _series_problem(B, n)
    K = ops.fill(ops.cumtrapz(dev(vol), dev(x), square=True))
    r = dev(y - mean)
    s2 = torch.full((B,), SIG2, device="cuda")
    ws = ops.MllWorkspace(B, n, True, K.device)
    out = ops.mll_step(K, r, s2, ws)[0].clone()
    assert int(ws.info.abs().sum()) == 0, (B, n, ws.info.tolist())
    rows = sorted({0, B - 1})
    _check_vs_oracle(K[rows].cpu().numpy(), y, mean, 1e-5, out.cpu().numpy(), ws.alpha.cpu().numpy(), rows)
    for _ in range(3):
        assert torch.equal(ops.mll_step(K, r, s2, ws)[0], out)
    res[f"{B}x{n}"] = "ok"
# fp64: the one-launch factorisation of a batch that is a multiple of 8, against LAPACK
B, n = 8, 1024
x, vol, y, mean = _series_problem(B, n)
K64 = ops.fill(ops.cumtrapz(dev(vol).double(), dev(x).double(), square=True)) + 0.5 * torch.eye(n, device="cuda", dtype=torch.float64)
f = ops.potrf(K64)
assert int(f.info.abs().sum()) == 0, f.info.tolist()
Lh = f.A[:, :n, :n].tril().clone()
Lr = torch.linalg.cholesky(K64.cpu())
assert float((Lh.cpu() - Lr).abs().max()) <= 1e-11 * float(Lr.abs().max()), "fp64 8x1024"
for _ in range(3 if (o[6] & 4) else 0):      # (the one-launch schedule has no atomics; the launch-per-column one adds K-slices atomically)
    assert torch.equal(ops.potrf(K64).A[:, :n, :n].tril(), Lh)
res["f64_8x1024"] = "ok"
print(json.dumps(res))
"""


def _run(extra, *argv):
    env = dict(os.environ)
    for k in ("VOLT_TUNE", "VOLT_FAKE_CUS", "VOLT_FAKE_XCCS", "HSA_CU_MASK", "VOLT_BATCH_XSKEW", "VOLT_BATCH_XDROP", "VOLT_BATCH_LOCAL"):
        env.pop(k, None)
    env.update(extra)
    out = subprocess.run([sys.executable, "-c", CODE, *argv], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-1500:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def test_planned_for_a_quarter_chip_on_the_real_one():
    r = _run({"VOLT_TUNE": "1", "VOLT_FAKE_CUS": "64", "VOLT_FAKE_XCCS": "2"})
    assert r["topology"] == [64, 2, 64, 80, 175, 175, 0]
    assert r["2x4096"] == r["8x399"] == r["1x1500"] == "ok"


def test_under_a_cu_mask_every_schedule_still_completes():
    """HSA_CU_MASK leaves the reported CU count alone (the guard does not see it) but takes three quarters of the CUs away
    from every queue: the default schedules -- including the one-launch steps, whose pieces wait for each other -- must still
    finish with info = 0 and oracle parity (dispatch-order hand-offs do not assume residency of later pieces)."""
    r = _run({"HSA_CU_MASK": "0:0-63"})
    assert r["2x4096"] == r["8x399"] == r["1x1500"] == "ok"
    assert r["8x2048"] == r["64x1024"] == r["f64_8x1024"] == "ok"          # LOCAL shapes: the pullers find their own XCD


@pytest.mark.parametrize("env", [
    {"VOLT_BATCH_XSKEW": "3"},                               # queue q runs on XCD q - 3: the map is nobody's assumption
    {"VOLT_BATCH_XDROP": "11"},                              # the pullers on XCDs 0, 1 and 3 leave at once: three orphan queues
    {"VOLT_BATCH_XDROP": "254"},                             # ONE XCD runs all eight queues
    {"VOLT_BATCH_XSKEW": "5", "VOLT_BATCH_XDROP": "36"},
    {"VOLT_BATCH_LOCAL": "0"},                               # one queue, the agent-scope protocol
], ids=["skew3", "drop_0_1_3", "one_xcd_left", "skew5_drop_2_5", "one_queue"])
def test_local_handoffs_do_not_depend_on_placement(env):
    """a forced 'wrong' map still gives oracle parity, bitwise repeatable -- every piece of a matrix runs
    under the L2 of the XCD that CLAIMED its queue, whichever that is."""
    r = _run({"VOLT_TUNE": "1", **env}, "local")
    assert r["8x2048"] == r["64x1024"] == r["16x1536"] == r["f64_8x1024"] == "ok"
