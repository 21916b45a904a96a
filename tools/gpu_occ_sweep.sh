#!/bin/bash
# 16-row-tile step kernel at 5 / 6 / 8 workgroups per CU (rebuilds the library on the box with -DGN_RPG1_OCC=k)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R && mkdir -p gpurun_out
for occ in ${OCCS:-5 6 8}; do
  GNODE_EXTRA_FLAGS="-DGN_RPG1_OCC=$occ" python gn-ode-sir_amd/gnode/build.py --force > gpurun_out/build_occ.log 2>&1 || { tail gpurun_out/build_occ.log; exit 1; }
  for s in 1 8; do
    echo -n "occ=$occ samples=$s  "
    GNODE_RPG=1 timeout -k 10 200 python bench.py --samples $s --chunk $s --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],3), 'step_us', round(d['roofline']['avg_launch_us'],1))" || exit 1
  done
done
