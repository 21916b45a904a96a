#!/bin/bash
# One GPU round: tests, bench, rocprof kernel trace.  Stops at the first step that hangs.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 700 python -m pytest tests -m gpu -x -q -s > gpurun_out/test.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -n 25 gpurun_out/test.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 500 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 $BENCH_ARGS > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?
echo "bench rc=$rc"; cat gpurun_out/bench.json; tail -n 5 gpurun_out/bench.err
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $R/gpurun_out/prof.log 2>&1; rc=$?
echo "rocprof rc=$rc"; tail -n 3 $R/gpurun_out/prof.log
find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -n 12
