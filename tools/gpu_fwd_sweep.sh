#!/bin/bash
# A/B of compile-time variants of the forward: bench.py at 1 and 8 samples + mid-size forward timings
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R && mkdir -p gpurun_out
IFS=';' read -ra VS <<< "${VARIANTS:--DGN_FWD_NB=4;-DGN_FWD_NB=8}"
for v in "${VS[@]}"; do
  GNODE_EXTRA_FLAGS="$v" python gn-ode-sir_amd/gnode/build.py --force > gpurun_out/build_occ.log 2>&1 || { tail gpurun_out/build_occ.log; exit 1; }
  echo "== $v"
  for s in 1 8; do
    echo -n "   samples=$s  "
    timeout -k 10 200 python bench.py --samples $s --chunk $s --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],3), 'step_us', round(d['roofline']['avg_launch_us'],1))" || exit 1
  done
  timeout -k 10 200 python tools/bench_configs.py mid 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ', d['case'][:28], 'fwd', round(d['forward_ms'],3), 'train', round(d['train_step_ms'],3))" || exit 1
done
