#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/prof_serialpmc -o ser -- python3 $ROOT/scripts/profile_grid.py --distinct 8 --points 2000000 --tile 64 --steps 1 --error-bound 0 > $OUT/prof_serialpmc.log 2>&1
echo rc=$?
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/prof_serialpmc/ser_counter_collection.csv")))
agg = collections.defaultdict(list)
for r in rows:
    if "k_grid_serial" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items(): print(k, v[-1])
PY
