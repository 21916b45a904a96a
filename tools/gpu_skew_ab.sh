#!/bin/bash
# A/B of compile-time constants on the degree-skew benchmark (tools/bench_skew.py): gpu_skew_ab.sh "<flags A>" "<flags B>" ...
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R && mkdir -p gpurun_out
for flags in "$@"; do
  echo "=== $flags"
  GNODE_EXTRA_FLAGS="$flags" python gn-ode-sir_amd/gnode/build.py --force > gpurun_out/build_ab.log 2>&1 || { tail gpurun_out/build_ab.log; exit 1; }
  timeout -k 10 300 python tools/bench_skew.py 2>/dev/null | python -c "
import sys, json
print(' '.join('%s/n%d/H%d:%.1f' % (d['graph'][:2], d['n'] // 1000, d['H'], d['us_per_step']) for d in map(json.loads, filter(lambda l: l.startswith('{'), sys.stdin))))"
done
GNODE_EXTRA_FLAGS="" python gn-ode-sir_amd/gnode/build.py --force > gpurun_out/build_ab.log 2>&1
