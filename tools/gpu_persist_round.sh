#!/bin/bash
# Evidence for the persistent one-launch kernels: bench lines, rocprofv3 kernel stats of the same script, per-phase stamps.
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
tag=${1:-r03}
OUT=$R/gpurun_out/$tag; mkdir -p $OUT; cd $R
timeout -k 10 300 python tools/bench_persist.py 2>/dev/null | grep '^{' > $OUT/${tag}_bench_persist.jsonl || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_persist
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_persist -- python3 $R/tools/bench_persist.py > $OUT/prof_persist.log 2>&1 || { tail -5 $OUT/prof_persist.log; exit 1; }
cp "$(find $OUT/prof_persist -name '*kernel_stats.csv' | head -1)" $OUT/${tag}_persist_kernel_stats.csv
rm -rf $OUT/prof_persist
cd $R
bash tools/gpu_persist_prof.sh 2>/dev/null | grep '^{' > $OUT/${tag}_persist_phases.jsonl
BENCH_ARGS=--tail bash tools/gpu_persist_prof.sh 2>/dev/null | grep '^{' >> $OUT/${tag}_persist_phases.jsonl
timeout -k 10 300 python tools/bench_small_h.py 2>/dev/null | grep '^{' > $OUT/${tag}_bench_small_h.jsonl
bash tools/gpu_small_h_prof.sh 2>/dev/null | grep '^{' > $OUT/${tag}_small_h_phases.jsonl
head -12 $OUT/${tag}_persist_kernel_stats.csv | cut -c1-50,100-105 | head -3
wc -l $OUT/${tag}_*.jsonl
