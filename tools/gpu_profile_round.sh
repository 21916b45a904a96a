#!/bin/bash
# Profiles of one round, written straight into profiles/<tag>_* (copied back through gpurun_out/<tag>/):
#   <tag>_bench.json                 the default bench line (python bench.py)
#   <tag>_kernel_stats.csv           rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary`
#   <tag>_full_kernel_stats.csv      the same of the default command (train + Monte-Carlo legs included)
#   <tag>_pmc_step64.json            FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ passes (own runs, kernel-trace only), corrected as
#                                    MI355X_MICROARCH.md prescribes; also written to pmc_step64_latest.json (what bench.py quotes)
#   <tag>_pmc_sq.txt                 SQ / TCP / TCC counter passes of the step kernel
# usage: GNODE_TREE=<commit> bash tools/gpu_profile_round.sh r02
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
tag=${1:-r02}
OUT=$R/gpurun_out/$tag; mkdir -p $OUT; cd $R
timeout -k 10 500 python bench.py > $OUT/${tag}_bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
prof() {   # name, bench args...
  name=$1; shift
  rm -rf $OUT/prof_$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- python3 $R/bench.py "$@" > $OUT/prof_$name.log 2>&1 || { tail -5 $OUT/prof_$name.log; return 1; }
  f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/${tag}_${name}.csv
  rm -rf $OUT/prof_$name
  head -6 $OUT/${tag}_${name}.csv | cut -c1-160
}
prof kernel_stats --steps 5 --warmup 2 --no-cpu-baseline --no-secondary || exit 1
prof full_kernel_stats --steps 5 --warmup 2 --no-cpu-baseline || exit 1
pmc() {    # counters... -> prints averaged per-dispatch values of the step kernel as JSON on one line
  rm -rf $OUT/pmc_raw
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_raw -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; return 1; }
  f=$(find $OUT/pmc_raw -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections, json
acc = collections.defaultdict(float); n = collections.Counter(); name = None
for row in csv.DictReader(open(sys.argv[1])):
    if row["Kernel_Name"].startswith("void k_step64<"):
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1; name = row["Kernel_Name"].split("(")[0]
print(json.dumps({"kernel": name, "dispatches": max(n.values()) if n else 0, **{c: v / n[c] for c, v in acc.items()}}))
PY
  rm -rf $OUT/pmc_raw
}
: > $OUT/pmc_lines.txt
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  pmc $C | tee -a $OUT/pmc_lines.txt || exit 1
done
python3 - $OUT $tag "${GNODE_TREE:-unknown}" <<'PY'
import json, sys, datetime
out, tag, tree = sys.argv[1:4]
d = {}
for line in open(f"{out}/pmc_lines.txt"):
    d.update(json.loads(line))
fetch_kb, write_kb = d["FETCH_SIZE"], d["WRITE_SIZE"]
rec = {"command": "rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary (one pass per counter group)",
       "workload": "ER n=75000 nnz=1000000 H=64, 8 samples per launch", "kernel": d["kernel"], "dispatches_averaged": d["dispatches"],
       "collected": datetime.date.today().isoformat(), "tree": tree,
       "FETCH_SIZE_KB_avg_per_dispatch": fetch_kb, "WRITE_SIZE_KB_avg_per_dispatch": write_kb,
       "traffic_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
       "fabric_read_bytes_per_launch": d["TCC_EA0_RDREQ_sum"] * 128,
       "l2_read_hit_rate": 1.0 - d["TCC_EA0_RDREQ_sum"] / d["TCP_TCC_READ_REQ_sum"],
       "l1_miss_read_bytes_per_launch": d["TCP_TCC_READ_REQ_sum"] * 128,
       "tcp_pending_stall_frac_of_cu_cycles": None,
       "counters": {k: v for k, v in d.items() if k not in ("kernel", "dispatches")},
       "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 FETCH_SIZE counts 64 B per 128-B request on 16-B/lane reads (all reads of this "
                     "kernel but the int32 CSR); counters are L2-to-fabric, Infinity-Cache hits included.  fabric_read = TCC_EA0_RDREQ x 128 B (agrees with "
                     "2*FETCH_SIZE); l1_miss_read = TCP_TCC_READ_REQ x 128 B"}
json.dump(rec, open(f"{out}/{tag}_pmc_step64.json", "w"), indent=1)
print(json.dumps({k: rec[k] for k in ("traffic_bytes_per_launch", "fabric_read_bytes_per_launch", "l2_read_hit_rate", "l1_miss_read_bytes_per_launch")}))
PY
cp $OUT/pmc_lines.txt $OUT/${tag}_pmc_sq.txt
