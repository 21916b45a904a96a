#!/bin/bash
# one PMC pass over bench.py for the step kernel: bash tools/gpu_pmc_one.sh <tag> COUNTER...   (env vars pass through)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
tag=$1; shift
OUT=$R/gpurun_out/$tag; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/raw
timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/raw -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
f=$(find $OUT/raw -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee -a $OUT/counters.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][:48]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for k, d in acc.items():
    if "k_step64" in k:
        print(k, {c: f"{v / n[(k, c)]:.4e}" for c, v in d.items()})
PY
rm -rf $OUT/raw
