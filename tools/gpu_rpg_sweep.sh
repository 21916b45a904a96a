#!/bin/bash
# 16-row vs 32-row tiles of the step kernel over the number of samples per launch (75k graph)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R && mkdir -p gpurun_out
for s in ${SWEEP:-1 2 3 4 6}; do
  for rpg in 1 2; do
    echo -n "samples=$s rpg=$rpg  "
    GNODE_RPG=$rpg timeout -k 10 200 python bench.py --samples $s --chunk $s --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],3), 'step_us', round(d['roofline']['avg_launch_us'],1))" || exit 1
  done
done
