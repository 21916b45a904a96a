#!/bin/bash
# rocprofv3 kernel trace of a python profile target: bash tools/gpu_prof_any.sh tools/prof_multi.py [args]; prints the
# kernel summary and the timeline of the last occurrence of $ANCHOR (default k_encode)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_any
T=$1; shift
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_any -- python3 $R/$T "$@" > $R/gpurun_out/prof_any.log 2>&1 || { tail -5 $R/gpurun_out/prof_any.log; exit 1; }
f=$(find $R/gpurun_out/prof_any -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
python3 $R/tools/timeline.py $R/gpurun_out/prof_any ${ANCHOR:-k_encode} | head -${LINES_MAX:-60}
