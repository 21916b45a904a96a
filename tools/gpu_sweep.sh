#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R; mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline $BARGS 2>gpurun_out/sweep.err | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('value %.4e  ms/step %.2f  step_us %.1f  frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac']))
"; }
for E in 8 125000 250000 500000 1000000; do BARGS="--edges $E" run GNODE_RPG=2; done
for E in 8 500000; do BARGS="--edges $E" run GNODE_RPG=2 GNODE_BENCH_OUT=last; done
