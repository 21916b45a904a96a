#!/bin/bash
# per-phase breakdown of the persistent integration: diagnostic build with in-kernel stamps, then the product build again
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R && mkdir -p gpurun_out
GNODE_EXTRA_FLAGS="-DGN_PERS_PROF $EXTRA" python gn-ode-sir_amd/gnode/build.py --force > gpurun_out/build_prof.log 2>&1 || { tail gpurun_out/build_prof.log; exit 1; }
timeout -k 10 300 python tools/bench_persist.py --prof $BENCH_ARGS > gpurun_out/bench_persist_prof.log 2>&1; rc=$?
cat gpurun_out/bench_persist_prof.log
GNODE_EXTRA_FLAGS="$EXTRA" python gn-ode-sir_amd/gnode/build.py --force > gpurun_out/build_prof.log 2>&1
exit $rc
