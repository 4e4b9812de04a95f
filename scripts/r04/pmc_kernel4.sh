#!/bin/bash
# pmc_kernel4.sh KERNEL_SUBSTRING SCRIPT [ARGS...]: SQ counters of the launches whose name contains KERNEL_SUBSTRING
# while `python3 SCRIPT ARGS` runs, per wave of the kernel's last launch (passes of at most eight counters, each its
# own run with --kernel-trace only, as gpurun asks).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
KERNEL=$1; shift
TAG=${PMC_TAG:-pmc4}
cd /tmp && export TMPDIR=/tmp
for pass in a b c d; do
  case $pass in
    a) C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY";;
    b) C="SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU";;
    c) C="SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_EXP_GDS";;
    d) C="SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH";;
  esac
  rm -rf $OUT/prof_${TAG}_$pass
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/prof_${TAG}_$pass -o run -- python3 $ROOT/"$1" "${@:2}" > $OUT/prof_${TAG}_$pass.log 2>&1
  python3 - <<PY
import csv, collections, glob
files = glob.glob("$OUT/prof_${TAG}_$pass/**/run_counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(files[0]))) if files else []
agg = collections.defaultdict(list)
for r in rows:
    if "$KERNEL" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
if agg:
    w = agg["SQ_WAVES"][-1]
    print("$KERNEL pass $pass: waves", w, "launches", len(agg["SQ_WAVES"]), {k: round(v[-1] / w, 1) for k, v in agg.items() if k != "SQ_WAVES"})
else:
    print("$KERNEL pass $pass: no counters", open("$OUT/prof_${TAG}_$pass.log").read()[-600:])
PY
  find $OUT/prof_${TAG}_$pass -name "*.csv" -size +20M -delete
done
