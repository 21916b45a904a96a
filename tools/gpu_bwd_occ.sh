#!/bin/bash
# fused backward with 16-row tiles at 3 / 4 workgroups per CU, and the 32-row form, on a big and two mid-size graphs
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R && mkdir -p gpurun_out
run() {
  bash tools/gpu_prof_train.sh 75000 500000 4 64 30 | grep -m1 k_bwd_fused64 || return 1
  timeout -k 10 200 python tools/bench_configs.py mid 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ', d['case'][:28], 'fwd', round(d['forward_ms'],3), 'train', round(d['train_step_ms'],3))"
}
for occ in ${OCCS:-3 4}; do
  GNODE_EXTRA_FLAGS="-DGN_BWD_RPG1_OCC=$occ" python gn-ode-sir_amd/gnode/build.py --force > gpurun_out/build_occ.log 2>&1 || { tail gpurun_out/build_occ.log; exit 1; }
  echo "== 16-row tiles, $occ workgroups per CU"; run || exit 1
done
echo "== 32-row tiles, 3 workgroups per CU"; GNODE_BWD_RPG=2 run
