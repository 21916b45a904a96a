#!/bin/bash
# PMC passes over one training shape (tools/train_75k.py n m B reps) for the training step kernel and the two backward
# interval kernels: per-dispatch averages, corrected as tools/gpu_profile_round.sh does for the inference step kernel.
#   usage: bash tools/gpu_pmc_train.sh <tag>        -> gpurun_out/<tag>/<tag>_pmc_train.json
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
tag=${1:-pmc_train}
OUT=$R/gpurun_out/$tag; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
: > $OUT/lines.txt
pass() {
  rm -rf $OUT/raw
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/raw -- python3 $R/tools/train_75k.py ${TRAIN_ARGS:-75000 500000 4 1} > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; return 1; }
  f=$(find $OUT/raw -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0]
    if k.startswith("void k_bwd_kept64<") or k.startswith("void k_step64<false"):
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for k, d in acc.items():
    print(json.dumps({"kernel": k, **{c: v / n[(k, c)] for c, v in d.items()}}))
PY
  rm -rf $OUT/raw
}
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  pass $C | tee -a $OUT/lines.txt || exit 1
done
python3 - $OUT $tag "${GNODE_TREE:-unknown}" <<'PY'
import json, sys, datetime, collections
out, tag, tree = sys.argv[1:4]
d = collections.defaultdict(dict)
for line in open(f"{out}/lines.txt"):
    r = json.loads(line); d[r.pop("kernel")].update(r)
rec = {"command": "rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python3 tools/train_75k.py 75000 500000 4 1 (one pass per counter group)",
       "workload": "training step, ER n=75000 nnz=1000000 H=64, 4 samples, 59 Euler steps, fused subsample (30 output grid points)",
       "collected": datetime.date.today().isoformat(), "tree": tree, "kernels": {},
       "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request on 16-B/lane reads); "
                     "L2-to-fabric, Infinity-Cache hits included; fabric_read = TCC_EA0_RDREQ x 128 B"}
for k, c in d.items():
    rec["kernels"][k] = {"traffic_bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
                         "fabric_read_bytes_per_launch": c["TCC_EA0_RDREQ_sum"] * 128,
                         "l2_read_hit_rate": 1.0 - c["TCC_EA0_RDREQ_sum"] / c["TCP_TCC_READ_REQ_sum"],
                         "counters": c}
json.dump(rec, open(f"{out}/{tag}_pmc_train.json", "w"), indent=1)
for k, v in rec["kernels"].items():
    print(k[:40], {a: (round(b, 4) if isinstance(b, float) and b < 10 else b) for a, b in v.items() if a != "counters"})
PY
