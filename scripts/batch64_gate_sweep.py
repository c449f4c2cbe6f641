"""Where the fp64 one-launch step (csrc/batch64_step.hip) beats chol64.hip's launch-per-column schedules: the same shapes
timed in two processes (the knobs are read once), VOLT_BATCH64=0 and VOLT_BATCH64=2, potrf and the gradient step.
    python scripts/batch64_gate_sweep.py [out.txt]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = [f"{B}x{N}" for N in (512, 1024, 2048, 3072, 4096) for B in (1, 2, 4, 8, 12, 16, 24, 32)
          if not (N >= 3072 and B > 24)]
if os.environ.get("SHAPES"):
    shapes = os.environ["SHAPES"].split(",")
res = {}
for mode in ("0", "2"):
    env = dict(os.environ, VOLT_TUNE="1", VOLT_BATCH64=mode)
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "batch64_check.py"), "--noref", *shapes], env=env,
                         capture_output=True, text=True, timeout=1500).stdout
    for line in out.splitlines():
        if line.startswith("{"):
            r = json.loads(line)
            res.setdefault(r["shape"], {})[mode] = r
lines = [f"{'shape':>9s} {'tiles':>6s} | potrf ms old -> one launch (ratio) | step ms old -> one launch (ratio)"]
for sh in shapes:
    if sh not in res or len(res[sh]) < 2: continue
    a, b = res[sh]["0"], res[sh]["2"]
    B, N = map(int, sh.split("x")); n = (N + 127) // 128
    lines.append(f"{sh:>9s} {B * (n + 1):6d} | {a['potrf_ms']:8.3f} -> {b['potrf_ms']:8.3f} ({a['potrf_ms'] / b['potrf_ms']:5.2f}) | "
                 f"{a['step_ms']:8.3f} -> {b['step_ms']:8.3f} ({a['step_ms'] / b['step_ms']:5.2f})")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt + "\n")
