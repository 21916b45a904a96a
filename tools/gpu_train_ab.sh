#!/bin/bash
# A/B helper: backward + trainer GPU tests, then the train legs of the bench line under rocprofv3 --stats (per-kernel
# averages of the step and interval kernels).  Output under gpurun_out/ab/.
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
OUT=$R/gpurun_out/ab; mkdir -p $OUT; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_abi.py -x -q > $OUT/tests.log 2>&1 || { tail -15 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/tools/train_75k.py > $OUT/prof.log 2>&1 || { tail -5 $OUT/prof.log; exit 1; }
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/train_kernel_stats.csv; rm -rf $OUT/prof
python3 - $OUT/train_kernel_stats.csv $OUT/prof.log <<'PY'
import csv, sys, json
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "avg_us %9.1f" % (float(r["AverageNs"]) / 1e3), "pct", r["Percentage"])
d = [json.loads(l) for l in open(sys.argv[2]) if l.startswith("{")][-1]
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if "bytes" not in k and "frac" not in k})
PY
