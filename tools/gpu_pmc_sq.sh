#!/bin/bash
# SQ counter pass for the step kernel (own run, kernel-trace only).
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_sq
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_sq.log 2>&1 || { tail -5 $R/gpurun_out/pmc_sq.log; exit 1; }
f=$(find $R/gpurun_out/pmc_sq -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][:40]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, d in acc.items():
    if "k_step64" in k or "k_mlp64" in k:
        print(k, {c: f"{v:.3e}" for c, v in d.items()})
PY
