#!/bin/bash
# PMC passes over the training profile target (tools/prof_train.py n m B H maxTime) for the backward interval kernel
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
OUT=$R/gpurun_out/${1:-pmc_train}; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
shift
pass() {
  rm -rf $OUT/raw
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/raw -- python3 $R/tools/prof_train.py ${TRAIN_ARGS:-75000 500000 4 64 30} > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; return 1; }
  f=$(find $OUT/raw -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][:48]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for k, d in acc.items():
    if "k_bwd_fused64<3, 1, false>" in k or "k_step64<false>" in k:
        print(k, {c: f"{v / n[(k, c)]:.4e}" for c, v in d.items()})
PY
  rm -rf $OUT/raw
}
pass SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES | tee $OUT/a.txt
pass TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum | tee $OUT/b.txt
pass TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum | tee $OUT/c.txt
pass SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE | tee $OUT/d.txt
