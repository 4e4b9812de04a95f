#!/bin/bash
# pmc_kernel.sh KERNEL_SUBSTRING SCRIPT [ARGS...]: SQ counters of the launches whose name contains
# KERNEL_SUBSTRING while `python3 SCRIPT ARGS` runs, per wave (two passes of eight counters).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
KERNEL=$1; shift
cd /tmp && export TMPDIR=/tmp
for pass in a b; do
  if [ $pass = a ]; then C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY";
  else C="SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU"; fi
  rm -rf $OUT/prof_pmc_$pass
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/prof_pmc_$pass -o run -- python3 $ROOT/"$1" "${@:2}" > $OUT/prof_pmc_$pass.log 2>&1
  python3 - <<PY
import csv, collections, glob
files = glob.glob("$OUT/prof_pmc_$pass/**/run_counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(files[0]))) if files else []
agg = collections.defaultdict(list)
for r in rows:
    if "$KERNEL" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
if agg:
    w = agg["SQ_WAVES"][-1]
    print("$KERNEL pass $pass: waves", w, "launches", len(agg["SQ_WAVES"]), {k: round(v[-1] / w, 1) for k, v in agg.items() if k != "SQ_WAVES"})
else:
    print("$KERNEL pass $pass: no counters", open("$OUT/prof_pmc_$pass.log").read()[-600:])
PY
done
