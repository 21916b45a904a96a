#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_train
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -- python3 $R/tools/prof_train.py "$@" > $R/gpurun_out/prof_train.log 2>&1
f=$(find $R/gpurun_out/prof_train -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
