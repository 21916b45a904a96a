#!/bin/bash
# A/B of compile-time constants of the small-hidden persistent kernels: gpu_small_h_ab.sh "<flags A>" "<flags B>" ... (2 repeats each)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R && mkdir -p gpurun_out
for flags in "$@"; do
  echo "=== $flags"
  GNODE_EXTRA_FLAGS="$flags" python gn-ode-sir_amd/gnode/build.py --force > gpurun_out/build_ab.log 2>&1 || { tail gpurun_out/build_ab.log; exit 1; }
  for rep in 1 2; do
    timeout -k 10 200 python tools/bench_small_h.py 2>/dev/null | python -c "
import sys, json
print(' '.join('%s:%.4f/%.4f' % (d['case'][:5], d['persist_fwd_ms'], d['persist_bwd_ms']) for d in map(json.loads, filter(lambda l: l.startswith('{'), sys.stdin)) if d['path'] == 3))"
  done
done
GNODE_EXTRA_FLAGS="" python gn-ode-sir_amd/gnode/build.py --force > gpurun_out/build_ab.log 2>&1
