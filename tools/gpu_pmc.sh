#!/bin/bash
# PMC passes for the L2<->fabric traffic of the step kernel (separate runs, kernel-trace only), summarised
# into gpurun_out/pmc_step64.json with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md.
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$C.log 2>&1 || { tail -3 $R/gpurun_out/pmc_$C.log; exit 1; }
done
python3 - $R <<'PY'
import csv, glob, json, sys
R = sys.argv[1]
out = {"command": "rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline (one pass per counter)",
       "workload": "ER n=75000 nnz=1000000 H=64, 8 samples per launch"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob(f"{R}/gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True))[-1]
    v = n = 0; name = None
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == c and row["Kernel_Name"].startswith("void k_step64<"):
            v += float(row["Counter_Value"]); n += 1; name = row["Kernel_Name"].split("(")[0]
    out["kernel"] = name
    out[c + "_KB_avg_per_dispatch"] = v / max(n, 1)
    out[c + "_dispatches"] = n
out["traffic_bytes_per_launch"] = (2 * out["FETCH_SIZE_KB_avg_per_dispatch"] + out["WRITE_SIZE_KB_avg_per_dispatch"]) * 1024
out["correction"] = ("traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 FETCH_SIZE counts 64 B per 128-B request on 16-B/lane reads "
                     "(all reads of this kernel but the int32 CSR); counters are L2-to-fabric, Infinity-Cache hits included")
json.dump(out, open(f"{R}/gpurun_out/pmc_step64.json", "w"), indent=1)
print(json.dumps(out))
PY
