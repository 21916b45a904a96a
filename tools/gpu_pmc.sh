#!/bin/bash
# PMC passes for HBM-side traffic of the step kernel (separate runs, kernel-trace only).
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$C.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1)
  python3 - "$f" $C <<'PY'
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(f)):
    if row.get("Counter_Name") != c: continue
    k = row["Kernel_Name"].split("(")[0][:60]
    acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:6]:
    print(f"{c} kernel={k} dispatches={n} avg_per_dispatch={v/n:.1f}")
PY
done
