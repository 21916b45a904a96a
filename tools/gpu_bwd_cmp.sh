#!/bin/bash
# backward tests, then the training profile on the 75k graph (B=4) for the fused backward at 2 and 3 workgroups per CU
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R && mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_backward.py tests/test_gpu_trainer.py -x -q > gpurun_out/bwd_test.log 2>&1; rc=$?
tail -n 6 gpurun_out/bwd_test.log
if [ $rc -ne 0 ]; then exit $rc; fi
for occ in 2 3; do
  echo "== GNODE_BWD_OCC=$occ"
  GNODE_BWD_OCC=$occ bash tools/gpu_prof_train.sh 75000 500000 4 64 30 | head -5 || exit 1
done
echo "== unfused"
GNODE_BWD_FUSE=0 bash tools/gpu_prof_train.sh 75000 500000 4 64 30 | head -5
