#!/bin/bash
# usage: pmc_fit.sh <variant lib name in scripts/ab> <series>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export MDB_HIP_LIBRARY=$ROOT/scripts/ab/${1}_libmdb_hip.so
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/prof_fitpmc_$1 -o fit -- python3 $ROOT/scripts/profile_fit.py --series ${2:-200} --points 10000000 > $OUT/prof_fitpmc_$1.log 2>&1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/prof_fitpmc_$1/fit_counter_collection.csv")))
agg = collections.defaultdict(list)
for r in rows:
    if "k_fit_models" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = agg["SQ_WAVES"][-1]
print("$1", "waves", w, {k: round(v[-1] / w / 65536, 1) for k, v in agg.items() if k != "SQ_WAVES"})
PY
