"""Host logic of the balanced launch schedule (csrc/sched.h) through the C-ABI's volt_sched_describe: no GPU needed.

Every tile of a launch must be covered exactly once, the K-slices of a cut tile must partition its K range in order,
the diagonal blocks must lead the grid (the panel tiles of the same launch poll their flags), and the pieces behind them
must come longest first (the dispatcher then does list scheduling in LPT order)."""
import ctypes as C

import numpy as np
import pytest

from volt_amd import _lib

DIAG, LOOKAHEAD, PANEL, TRTRI, TRTRI_DIAG = range(5)


def describe(B, n, has_y, k, G=256, S=4, frac=0.6):
    L = _lib.lib()
    items = np.zeros((1 << 15, 4), dtype=np.int32)
    loads = np.zeros(G, dtype=np.float32)
    c = L.volt_sched_describe(B, n, int(has_y), k, G, S, C.c_float(frac), items.ctypes.data, items.shape[0], loads.ctypes.data)
    assert c >= 0, c
    it = items[:c]
    return dict(kind=it[:, 0] & 7, b=it[:, 0] >> 3, idx=it[:, 1], sl=it[:, 2] & 255, nsl=(it[:, 2] >> 8) & 255,
                tile=it[:, 2] >> 16, b0=it[:, 3] & 0xffff, b1=it[:, 3] >> 16), loads


@pytest.mark.parametrize("B,n,has_y", [(8, 32, True), (12, 32, True), (5, 7, False), (9, 3, True), (24, 16, True)])
def test_every_tile_once_and_slices_partition(B, n, has_y):
    for k in range(n + (1 if has_y else 0)):
        d, loads = describe(B, n, has_y, k, S=max(1, min(4, 64 // B)))
        itri = (k - 1 if k < n else n - 1) if has_y else -1
        want = {}
        if k < n:
            for b in range(B):
                want[(DIAG, b, k)] = 0
                if 1 <= k < n - 1:
                    want[(LOOKAHEAD, b, k + 1)] = k
                for i in range(k + 1, n):
                    want[(PANEL, b, i)] = k
        for j in range(itri + 1):
            for b in range(B):
                want[(TRTRI_DIAG if j == itri else TRTRI, b, j)] = 0 if j == itri else itri - j
        got = {}
        for i in range(len(d["kind"])):
            key = (int(d["kind"][i]), int(d["b"][i]), int(d["idx"][i]))
            got.setdefault(key, []).append(i)
        assert set(got) == set(want), (k, set(got) ^ set(want))
        tiles_seen = {}
        for key, idxs in got.items():
            kb = want[key]
            if key[0] in (DIAG, TRTRI_DIAG):
                assert len(idxs) == 1
                continue
            nsl = int(d["nsl"][idxs[0]])
            assert len(idxs) == nsl and 1 <= nsl <= 4
            order = sorted(idxs, key=lambda i: d["sl"][i])
            assert [int(d["sl"][i]) for i in order] == list(range(nsl))
            assert int(d["b0"][order[0]]) == 0 and int(d["b1"][order[-1]]) == kb
            for a, c in zip(order, order[1:]):
                assert int(d["b1"][a]) == int(d["b0"][c])
            tile = {int(d["tile"][i]) for i in idxs}
            assert len(tile) == 1                                  # one slab / counter slot per tile ...
            t = tile.pop()
            assert 0 <= t < B * (n + 1) and t not in tiles_seen    # ... inside the launch's counter row, not shared
            tiles_seen[t] = key
        # grid order: the diagonal blocks first
        nd = B if k < n else 0
        assert (d["kind"][:nd] == DIAG).all() and (d["kind"][nd:] != DIAG).all()
        assert loads.max() > 0


def test_longest_first_and_balance():
    B, n, k = 8, 32, 31
    d, loads = describe(B, n, True, k)
    length = (d["b1"] - d["b0"])[B:]
    # falling cost: K length never grows by more than the per-slice constants allow (ties keep enumeration order)
    assert (np.diff(length.astype(int)) <= 1).all()
    # the schedule this is for: the longest uncut tile would be k + phase 2 ~ 32 blocks; cut, no slot carries more than ~2/3 of it
    assert loads.max() < 22 and loads.max() < 1.25 * loads.mean()


def test_bad_arguments():
    L = _lib.lib()
    assert L.volt_sched_describe(0, 4, 1, 0, 256, 4, C.c_float(0.6), None, 0, None) == -1
    assert L.volt_sched_describe(4, 4, 0, 4, 256, 4, C.c_float(0.6), None, 0, None) == -1     # k == n needs the inverse
    assert L.volt_sched_describe(8, 32, 1, 16, 256, 4, C.c_float(0.6), None, 1, None) == -2


def test_topology_guard_scales_the_gates_and_switches_the_one_launch_steps_off():
    """Every schedule gate was measured on a full MI355X (256 CUs in 8 XCDs).  The library asks the device once
    (multiprocessor count, hipDeviceAttributeNumberOfXccs); on anything else the slot-count gates scale with the CU count and
    the one-launch steps -- tuned for the full chip, the batched one relying on one XCD per matrix -- are off, so such a
    device runs the launch-per-column schedules.  Planned here for a faked 64-CU / 2-XCD device (a CPX-
    like partition) in a subprocess (the knobs are read once per process); no GPU."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes as C, json\nfrom volt_amd import _lib\nL = _lib.lib()\no = (C.c_int * 7)()\nL.volt_topology_describe(o)\n"
            "print(json.dumps({'t': list(o), 'small': L.volt_mll_workspace_bytes(8, 399, 1), 'long': L.volt_mll_workspace_bytes(1, 4096, 1),"
            " 'b64': L.volt_mll_workspace_bytes(64, 4096, 1), 'p65': L.volt_potrf_workspace_bytes(65, 4096)}))")

    def run(env_extra):
        env = dict(os.environ)
        for k in ("VOLT_TUNE", "VOLT_FAKE_CUS", "VOLT_FAKE_XCCS"):
            env.pop(k, None)
        env.update(env_extra)
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-800:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    full = run({})
    assert full["t"] == [256, 8, 256, 320, 700, 700, 7]
    small = run({"VOLT_TUNE": "1", "VOLT_FAKE_CUS": "64", "VOLT_FAKE_XCCS": "2"})
    assert small["t"] == [64, 2, 64, 80, 175, 175, 0]
    # no state of a one-launch step is reserved any more: the workspaces shrink to what the launch-per-column path needs
    assert small["small"] < full["small"] and small["long"] < full["long"] and small["b64"] < full["b64"]
    assert small["p65"] == 0 and full["p65"] > 0
    # a fake without VOLT_TUNE=1 is ignored (a deployment reads no environment)
    assert run({"VOLT_FAKE_CUS": "64"})["t"] == full["t"]
