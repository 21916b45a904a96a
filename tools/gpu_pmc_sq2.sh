#!/bin/bash
# Two SQ counter passes for the step kernel (own runs, kernel-trace only) + the list of counters the box offers.
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
OUT=$R/gpurun_out/${1:-pmc_sq2}
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*" | sort -u > $OUT/counters.txt
pass() {
  tag=$1; shift
  rm -rf $OUT/$tag
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$tag -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/$tag.log 2>&1 || { tail -5 $OUT/$tag.log; return 1; }
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][:48]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for k, d in acc.items():
    if "k_step64" in k:
        print(k, {c: f"{v / n[(k, c)]:.4e}" for c, v in d.items()})
PY
  rm -rf $OUT/$tag
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS | tee $OUT/pass_a.txt
pass b SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM | tee $OUT/pass_b.txt
pass c SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES | tee $OUT/pass_c.txt
