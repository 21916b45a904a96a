#!/bin/bash
# A/B of the Monte-Carlo kernel's compile-time variants: gpu_sir_ab.sh "<flags A>" "<flags B>" ...
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R && mkdir -p gpurun_out
for flags in "$@"; do
  echo "=== $flags"
  GNODE_EXTRA_FLAGS="$flags" python gn-ode-sir_amd/gnode/build.py --force > gpurun_out/build_ab.log 2>&1 || { tail gpurun_out/build_ab.log; exit 1; }
  timeout -k 10 300 python tools/bench_sir.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-16s b=%.2f g=%.2f  %.2f ms  %.3g traj-steps/s  exact=%s  %s' % (d['case'], d['beta'], d['gamma'], d['gpu_s']*1e3, d['gpu_traj_steps_per_s'], d['bit_exact_vs_oracle'], d['counted']))"
done
GNODE_EXTRA_FLAGS="" python gn-ode-sir_amd/gnode/build.py --force > gpurun_out/build_ab.log 2>&1
