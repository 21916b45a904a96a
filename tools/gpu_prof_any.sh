#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python script: tools/gpu_prof_any.sh <script.py> [args]
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_any
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_any -- python3 "$R/$1" "${@:2}" > $R/gpurun_out/prof_any.log 2>&1
grep -v amdgpu $R/gpurun_out/prof_any.log | tail -2
f=$(find $R/gpurun_out/prof_any -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
