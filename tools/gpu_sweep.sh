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
run GNODE_BENCH_OUT=all
run GNODE_BENCH_OUT=sub
run GNODE_BENCH_OUT=last
run GNODE_BENCH_OUT=all GNODE_FUSE=0
