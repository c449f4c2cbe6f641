"""Off the full chip: the library's gates were measured on 256 CUs in 8 XCDs; on any other device it
scales the slot gates and keeps to the launch-per-column schedules.  Two subprocesses (the knobs and the device are read
once per process) run 2 x 4096, 8 x 399 and 1 x 1500 -- shapes whose default schedules are the split-K launches, the
short-series one-launch step and the long-series one-launch step -- against the fp64 oracle:
  * planned for a faked 64-CU / 2-XCD device (VOLT_TUNE=1 VOLT_FAKE_CUS=64 VOLT_FAKE_XCCS=2) on the real one;
  * under HSA_CU_MASK (a quarter of the CUs), where the hand-offs of whatever schedule runs must still complete."""
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
for B, n in ((2, 4096), (8, 399), (1, 1500)):
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
print(json.dumps(res))
"""


def _run(extra):
    env = dict(os.environ)
    for k in ("VOLT_TUNE", "VOLT_FAKE_CUS", "VOLT_FAKE_XCCS", "HSA_CU_MASK"):
        env.pop(k, None)
    env.update(extra)
    out = subprocess.run([sys.executable, "-c", CODE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
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
