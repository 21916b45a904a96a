#!/bin/bash
# A/B of compile-time tuning constants on the 75k x 4 training step: bash tools/gpu_flag_ab.sh "<flags A>" "<flags B>" ...
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}; cd $R
for F in "$@"; do
  GNODE_EXTRA_FLAGS="$F" python gn-ode-sir_amd/gnode/build.py --force > /dev/null 2>&1 || { echo "build failed: $F"; exit 1; }
  python tools/train_75k.py 75000 500000 4 3 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('flags [$F]', 'step_ms %.2f' % d['ms_per_step'], 'fwd_us %.1f' % d['fwd_step_kernel_avg_us'], 'bwd_us %.1f' % d['bwd_interval_kernel_avg_us'])"
done
python gn-ode-sir_amd/gnode/build.py --force > /dev/null 2>&1
