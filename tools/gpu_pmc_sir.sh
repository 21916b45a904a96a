#!/bin/bash
# SQ counter passes of the Monte-Carlo kernel (k_sir_frontier) on configs[2]'s shape -> profiles/<tag>_pmc_sir.json
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
tag=${1:-r03}; shift
OUT=$R/gpurun_out/$tag; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/pmc_sir_lines.txt
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "GRBM_GUI_ACTIVE"; do
  rm -rf $OUT/pmc_raw
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_raw -- python3 $R/tools/prof_sir.py "$@" > $OUT/pmc_sir.log 2>&1 || { tail -5 $OUT/pmc_sir.log; exit 1; }
  f=$(find $OUT/pmc_raw -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY' | tee -a $OUT/pmc_sir_lines.txt
import csv, sys, collections, json
acc = collections.defaultdict(list); name = None
for row in csv.DictReader(open(sys.argv[1])):
    if "k_sir_frontier" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"])); name = row["Kernel_Name"].split("(")[0]
# the big launches only (the warm-up launch of 64 trajectories is ~150x smaller)
print(json.dumps({"kernel": name, **{c: sum(sorted(v)[-3:]) / 3 for c, v in acc.items()}}))
PY
  rm -rf $OUT/pmc_raw
done
python3 - $OUT $tag "$*" <<'PY'
import json, sys
out, tag, args = sys.argv[1:4]
d = {}
for line in open(f"{out}/pmc_sir_lines.txt"):
    d.update(json.loads(line))
wc = d.get("SQ_WAVE_CYCLES", 0.0) or 1.0
rec = {"command": "rocprofv3 --pmc <C> --kernel-trace -- python3 tools/prof_sir.py " + args, "workload": "ER n=7066 nnz=201472, 10 000 sims x T=20, 2 seeds",
       "kernel": d.get("kernel"), "counters_avg_per_dispatch": {k: v for k, v in d.items() if k != "kernel"},
       "wave_cycle_shares": {"wait_any (s_waitcnt / barrier)": d.get("SQ_WAIT_ANY", 0) / wc, "wait_inst_any (issue stall)": d.get("SQ_WAIT_INST_ANY", 0) / wc,
                             "active_inst_any": d.get("SQ_ACTIVE_INST_ANY", 0) / wc, "active_inst_valu": d.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                             "active_inst_lds": d.get("SQ_ACTIVE_INST_LDS", 0) / wc},
       "note": "SQ_WAVE_CYCLES etc. count quad-cycles summed over waves (MI355X_MICROARCH.md); shares are of wave lifetime"}
json.dump(rec, open(f"{out}/{tag}_pmc_sir.json", "w"), indent=1)
print(json.dumps(rec["wave_cycle_shares"]))
PY
