#!/bin/bash
# per-phase stamps of the small-hidden-size persistent forward (diagnostic build), then the product build again
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R && mkdir -p gpurun_out
GNODE_EXTRA_FLAGS="-DGN_PERS_PROF" python3 gn-ode-sir_amd/gnode/build.py --force > gpurun_out/build_prof.log 2>&1 || { tail -n 20 gpurun_out/build_prof.log; exit 1; }
timeout -k 10 300 python3 tools/bench_small_h.py --prof 2>/dev/null | tee gpurun_out/small_h_phases.jsonl
rc=$?
python3 gn-ode-sir_amd/gnode/build.py --force > gpurun_out/build_back.log 2>&1
exit $rc
