"""Randomised sweep of the MLL step through the small-batch / balanced schedules at sizes that reach the scheduled block
columns (N 2200..4100, odd sizes and ragged padding included), against the fp64 oracle on two series per case and against
the vendor's fp32 factorisation of the same matrices.  The cases and the gates live in tests/fuzz_sched_cases.py (pytest
runs a few of them: tests/test_gpu_lownoise.py).  Exits non-zero on a miss.
    python scripts/fuzz_sched.py [seed [cases]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import fuzz_sched_cases as fz

worst = fz.run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 14, verbose=True)
print("worst", {k: float(f"{v:.3g}") for k, v in worst.items()})
bad = fz.failures(worst)
print("gates", fz.GATE, "->", "ok" if not bad else f"MISSED {bad}")
sys.exit(1 if bad else 0)
