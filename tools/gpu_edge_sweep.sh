#!/bin/bash
# step-kernel launch time vs edge count at fixed n (75k x 8 samples): intercept = edge-free base, slope = gather cost
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
OUT=$R/gpurun_out/${1:-sweep}; mkdir -p $OUT; cd $R
for v in ${VARIANTS:-"GNODE_X=0"}; do
  for m in ${EDGES:-16 250000 500000 1000000}; do
    env $v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --edges $m > $OUT/b.json 2> $OUT/b.err || tail -3 $OUT/b.err
    python - "$OUT/b.json" "$v m=$m" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "launch_us %.1f" % d["roofline"]["avg_launch_us"], "ms/step %.3f" % d["ms_per_step"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
