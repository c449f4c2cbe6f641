"""Is the schedule the library picks for a shape the fastest of the alternatives it could have picked?  (VERDICT r03 weak
point 11: five step schedules behind hand-tuned (B, N) gates.)  For a handful of shapes on either side of the gates, time
the MLL+grad step with the frozen defaults and with each alternative forced through the experiment knobs (VOLT_TUNE=1 +
VOLT_*; one subprocess per variant -- the knobs are read once per process), and report default / best.
    python scripts/sched_choice_check.py [--json]        exit code 1 if the default is > 5 % behind the best alternative"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [   # shape, {label: env}
    # (round 5) the one-launch batched step is the default for the first eight; "launch per column" = the round-4 schedules
    ("64x4096", {"launch per column, 2 groups": {"VOLT_BATCH": "0"}, "launch per column, 1 group": {"VOLT_BATCH": "0", "VOLT_GROUPS": "1"}}),
    ("64x2048", {"launch per column, 2 groups": {"VOLT_BATCH": "0"}, "launch per column, 4 groups": {"VOLT_BATCH": "0", "VOLT_GROUPS": "4"}}),
    ("32x4096", {"launch per column": {"VOLT_BATCH": "0"}, "launch per column, balanced": {"VOLT_BATCH": "0", "VOLT_SCHED_MAXB": "32"}}),
    ("16x4096", {"launch per column, balanced": {"VOLT_BATCH": "0"}, "launch per column, plain": {"VOLT_BATCH": "0", "VOLT_SCHED": "0"}}),
    ("8x4096", {"launch per column, balanced": {"VOLT_BATCH": "0"}, "launch per column, plain": {"VOLT_BATCH": "0", "VOLT_SCHED": "0"}}),
    ("4x4096", {"launch per column, balanced": {"VOLT_BATCH": "0"}, "two workgroups per CU": {"VOLT_BATCH_SPREAD": "0"}}),
    ("2x4096", {"launch per column, split-K": {"VOLT_BATCH": "0"}, "two workgroups per CU": {"VOLT_BATCH_SPREAD": "0"}}),
    ("16x2048", {"launch per column": {"VOLT_BATCH": "0"}, "two workgroups per CU": {"VOLT_BATCH_SPREAD": "0"}}),
    ("4x2048", {"one launch (batched step)": {"VOLT_BATCH": "2"}}),
    ("1x4096", {"launch per column": {"VOLT_LONG": "0"}, "one-workgroup spine": {"VOLT_LONG_SPLIT": "0"}}),
    ("1x399", {"launch per column": {"VOLT_LONG": "0", "VOLT_SMALL_NMAX": "0"}, "short-series kernel": {"VOLT_LONG": "0"}}),
    ("8x399", {"launch per column": {"VOLT_SMALL_NMAX": "0"}, "batched step": {"VOLT_BATCH": "3"}}),
    ("64x399", {"launch per column": {"VOLT_SMALL_NMAX": "0"}}),
]


def run(shape, env_extra):
    env = dict(os.environ)
    env.pop("VOLT_TUNE", None)
    if env_extra:
        env.update(env_extra, VOLT_TUNE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "quick_step.py"), shape], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    for ln in out.stdout.splitlines():
        if "ms/step" in ln:
            return float(ln.split(":")[1].split("ms/step")[0])
    raise RuntimeError(out.stderr[-500:])


def main():
    rows, worst = [], 0.0
    for shape, alts in CASES:
        d = run(shape, None)
        res = {lab: run(shape, env) for lab, env in alts.items()}
        best = min([d] + list(res.values()))
        worst = max(worst, d / best)
        rows.append({"shape": shape, "default_ms": d, "alternatives_ms": res, "default_over_best": round(d / best, 4)})
        print(f"{shape:>8s}: default {d:8.4f} ms | " + " | ".join(f"{k} {v:.4f}" for k, v in res.items()) + f" | default/best {d / best:.3f}", flush=True)
    if "--json" in sys.argv:
        print(json.dumps({"rows": rows, "worst_default_over_best": worst}))
    return 0 if worst <= 1.05 else 1


if __name__ == "__main__":
    sys.exit(main())
