#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline $BARGS 2>gpurun_out/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('value %.4e  ms/step %.2f  step_us %.1f  mlp_us %.1f  frac %.3f chunk %s' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['node_mlp_avg_launch_us'], r['frac'], d['config']['samples_per_launch']))
"; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/test.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 2 gpurun_out/test.log
for i in 1 2; do
run GNODE_XQ=1
run GNODE_XQ=0
done
run GNODE_XQ=1 GNODE_CHUNK=4
run GNODE_XQ=0 GNODE_CHUNK=4
run GNODE_XQ=1 GNODE_CHUNK=1
run GNODE_XQ=0 GNODE_CHUNK=1
