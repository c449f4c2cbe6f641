#!/bin/bash
# Issue-side counters of rollout_bordered_kernel<1,false> at config 5's per-GPU share (8 series x 10^4 paths x 256 steps,
# N = 4096): what the kernel's roofline is made of.  One SQ pass (8 slots) + GRBM, hard timeout;
# summary -> gpurun_out/pmc_rollissue/rollout_pmc.json, carrying the hash of the library sources it was taken on: copy to
# profiles/rollout_pmc.json (bench.py refuses it for any other sources) and to profiles/r06/.
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_rollissue
cd /tmp
timeout -k 5 ${PMC_TIMEOUT:-300} rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $R/gpurun_out/pmc_rollissue/sq -o sq -- python $R/scripts/bench_rollouts.py > $R/gpurun_out/pmc_rollissue/sq.log 2>&1
echo "pass sq rc=$?"
timeout -k 5 ${PMC_TIMEOUT:-300} rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $R/gpurun_out/pmc_rollissue/sq2 -o sq2 -- python $R/scripts/bench_rollouts.py > $R/gpurun_out/pmc_rollissue/sq2.log 2>&1
echo "pass sq2 rc=$?"
cd $R
python - <<'PY'
import csv, datetime, glob, json, sys
from collections import defaultdict
sys.path.insert(0, ".")
from volt_amd.build import source_hash
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for f in glob.glob("gpurun_out/pmc_rollissue/sq*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "rollout_lane_kernel<0, true>" in k or "rollout_lane_kernel<0,true>" in k:
            agg["k"][row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
a = agg["k"]
disp = max(n.values()) if n else 0
per = {c: v / max(1, n[c]) for c, v in a.items()}
G, S, H = 8, 10000, 256
out = {"kernel": "rollout_lane_kernel<0, true>", "source_hash": source_hash(), "date": datetime.date.today().isoformat(), "workload": f"{G} series x {S} paths x {H} steps, N=4096",
       "shape": {"G": G, "S": S, "H": H, "n": 4096}, "dispatches": disp,
       "per_dispatch": per}
if "SQ_INSTS_VALU" in per:
    out["valu_wave_insts_per_sample_step"] = per["SQ_INSTS_VALU"] / (G * S * H)
    out["salu_insts_per_sample_step"] = per.get("SQ_INSTS_SALU", 0) / (G * S * H)
    w = per.get("SQ_WAVE_CYCLES", 0)
    if w:
        out["active_inst_valu_over_wave_cycles"] = per.get("SQ_ACTIVE_INST_VALU", 0) / w
        out["wait_inst_any_over_wave_cycles"] = per.get("SQ_WAIT_INST_ANY", 0) / w
        out["active_inst_any_over_wave_cycles"] = per.get("SQ_ACTIVE_INST_ANY", 0) / w
        out["wait_any_over_wave_cycles"] = per.get("SQ_WAIT_ANY", 0) / w
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_rollissue/rollout_pmc.json", "w"), indent=1)
PY
