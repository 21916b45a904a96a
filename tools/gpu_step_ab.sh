#!/bin/bash
# A/B of the step kernel on the benchmark workload + the forward parity tests; $1 = output tag
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
OUT=$R/gpurun_out/${1:-ab}; mkdir -p $OUT; cd $R
if [ -z "$SKIP_PARITY" ]; then python -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -x -q -k "forward or full or golden or independence or edge or tiny or skewed or randomized or backward or adam or train" > $OUT/parity.log 2>&1; echo "parity rc=$?" | tee -a $OUT/parity.log; fi
tail -3 $OUT/parity.log
for v in ${VARIANTS:-"GNODE_X=0"}; do
  env $v python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err || tail -3 $OUT/bench_$v.err
  python - "$OUT/bench_$v.json" "$v" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value %.4g" % d["value"], "ms/step %.3f" % d["ms_per_step"], "launch_us %.1f" % d["roofline"]["avg_launch_us"], "valid", d["config"]["outputs_valid"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
