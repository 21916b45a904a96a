#!/bin/bash
# A/B of compile-time variants: rebuild the library on the box once per flag set in VARIANTS (';'-separated), then
# run the training profile on the 75k graph (4 samples) and the mid-size forward / train-step timings.
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R && mkdir -p gpurun_out
IFS=';' read -ra VS <<< "${VARIANTS:--DGN_BWD_NB=2;-DGN_BWD_NB=4}"
for v in "${VS[@]}"; do
  GNODE_EXTRA_FLAGS="$v" python gn-ode-sir_amd/gnode/build.py --force > gpurun_out/build_occ.log 2>&1 || { tail gpurun_out/build_occ.log; exit 1; }
  echo "== $v"
  bash tools/gpu_prof_train.sh 75000 500000 4 64 30 | grep -m2 "k_bwd_fused64\|k_step64" || exit 1
  timeout -k 10 200 python tools/bench_configs.py mid 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ', d['case'][:28], 'fwd', round(d['forward_ms'],3), 'train', round(d['train_step_ms'],3))" || exit 1
done
