# HBM traffic of the one-launch step by piece order: FETCH_SIZE / WRITE_SIZE passes of `bench.py --profile-only 3`
set -x
cd $GRAFT_REPO_ROOT
export VOLT_TUNE=1
for o in ${ORDERS:-0 54 118 22}; do
  export VOLT_BATCH_ORDER=$o
  PMC_PASSES="fetch write" scripts/pmc.sh ord$o --profile-only 3 > /dev/null 2>&1
  python - $o <<'PY'
import csv, glob, sys
o = sys.argv[1]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = n = 0
    for f in glob.glob(f"gpurun_out/pmc_ord{o}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "batch_step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v += float(r["Counter_Value"]); n += 1
    tot[c] = (v / max(n, 1), n)
rd, wr = 2 * tot["FETCH_SIZE"][0] * 1024, tot["WRITE_SIZE"][0] * 1024
print(f"order {o}: read {rd/1e9:.2f} GB + write {wr/1e9:.2f} GB = {(rd+wr)/1e9:.2f} GB per launch ({tot['FETCH_SIZE'][1]} launches)", flush=True)
PY
done | tee gpurun_out/r06/order_traffic.txt
