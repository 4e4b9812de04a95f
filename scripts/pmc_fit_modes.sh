#!/bin/bash
# Instruction counters of k_fit_models per input point: mode 0 plain, 1 fast forms, 2 the lean kernel.
# usage: pmc_fit_modes.sh [series] ; writes gpurun_out/pmc_fit_<mode>.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
SERIES=${1:-1000}
cd /tmp && export TMPDIR=/tmp
for fast in 1 2; do
  for pass in a b; do
    if [ $pass = a ]; then C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY";
    else C="SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_INSTS_BRANCH"; fi
    if [ $fast = 2 ]; then export MDB_FIT_FAST=1 MDB_FIT_LEAN=1; else export MDB_FIT_FAST=$fast MDB_FIT_LEAN=0; fi; export MDB_FIT_PIECE_POINTS=1
    rm -rf $OUT/prof_fitpmc_$fast$pass
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/prof_fitpmc_$fast$pass -o fit -- python3 $ROOT/scripts/profile_fit.py --series $SERIES --points 10000000 > $OUT/prof_fitpmc_$fast$pass.log 2>&1
    python3 - <<PY
import csv, collections, glob
files = glob.glob("$OUT/prof_fitpmc_$fast$pass/**/fit_counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(files[0]))) if files else []
agg = collections.defaultdict(list)
for r in rows:
    if "k_fit_models" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
if agg:
    w = agg["SQ_WAVES"][-1]
    print("fast=$fast pass $pass waves", w, {k: round(v[-1] / w / 65536, 2) for k, v in agg.items() if k != "SQ_WAVES"})
else:
    print("fast=$fast pass $pass: no counters", open("$OUT/prof_fitpmc_$fast$pass.log").read()[-800:])
PY
  done
done
