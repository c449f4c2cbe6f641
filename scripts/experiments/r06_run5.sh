set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python bench.py --steps 20 --no-cpu-baseline > gpurun_out/r06/bench_try1.json 2> gpurun_out/r06/bench_try1.err; tail -3 gpurun_out/r06/bench_try1.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06/bench_try1.json").read().strip().splitlines()[-1])
print("value",d["value"],"ms",d["ms_per_step"],"roof",d["roofline"]["frac"] if d.get("roofline") else None)
print(json.dumps(d.get("next_rows"),indent=1)[:4000])
print({k:(v["ms_per_step"],v["frac"]) for k,v in (d.get("configs") or {}).items()})
print({k:d["rollouts"][k] for k in ("total_s","kernel_ms","roofline","roofline_absent_because")})
print(json.dumps(d.get("fp64"))[:1500])
PY
export BS=2,3,4,6,8,12,16,20,24 NS=1024,1536,2048,3072
VOLT_TUNE=1 VOLT_BATCH=0 python scripts/batch_gate_sweep.py gpurun_out/r06/gate_a.json > gpurun_out/r06/gate_a.log 2>&1
VOLT_TUNE=1 VOLT_BATCH=3 python scripts/batch_gate_sweep.py gpurun_out/r06/gate_b.json > gpurun_out/r06/gate_b.log 2>&1
python scripts/batch_gate_sweep.py --join gpurun_out/r06/gate_a.json gpurun_out/r06/gate_b.json | tee gpurun_out/r06/batch_gate_sweep.txt
