"""The schedule the library picks for a shape against the alternatives it could have picked (VERDICT r03 weak point 11,
r04 item 8): for shapes on either side of the (B, N) gates the frozen default must be within 8 % of the fastest variant that
the experiment knobs can force (scripts/sched_choice_check.py; one subprocess per variant, VOLT_TUNE=1).  The full table is
profiles/r05/sched_choice.txt; the test runs seven of the shapes.

This is a TIMING assertion inside the parity gate, so it is built not to turn the gate red on one noisy sample: a shape
that misses on the first pass is re-timed three times (default and the alternative that beat it) and judged on the MEDIANS;
and the file sorts last (test_gpu_zz_*), so under `-x` every correctness test has run before it."""
import os
import statistics
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_schedule_is_the_fastest_alternative():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import sched_choice_check as scc
    keep = {"64x4096", "64x2048", "16x4096", "8x4096", "1x4096", "1x399", "8x399"}
    for shape, alts in scc.CASES:
        if shape not in keep:
            continue
        d = scc.run(shape, None)
        res = {lab: scc.run(shape, env) for lab, env in alts.items()}
        lab, best = min(res.items(), key=lambda kv: kv[1])
        if d <= 1.08 * best:
            continue
        # a miss: re-time this pair before believing it
        d3 = statistics.median([d] + [scc.run(shape, None) for _ in range(3)])
        b3 = statistics.median([best] + [scc.run(shape, alts[lab]) for _ in range(3)])
        assert d3 <= 1.08 * b3, (shape, lab, d3, b3, res)
