#!/bin/bash
# Memory-side counters of k_grid_mv_pieces on the mixed series (scripts/r04/mixed_grid.py): requests of the vector L1s to
# the L2, the L2's hits and misses, bytes fetched from HBM - one rocprofv3 --pmc pass each (with --kernel-trace only).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "TCP_TCC_READ_REQ_sum" "TCC_REQ_sum" "TCC_EA0_RDREQ_sum" "SQ_INSTS_VMEM_RD SQ_WAVES"; do
  tag=$(echo $C | tr ' ' '_')
  rm -rf $OUT/prof_mvmem_$tag
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/prof_mvmem_$tag -o run -- python3 $ROOT/scripts/r04/mixed_grid.py > $OUT/prof_mvmem_$tag.log 2>&1
  python3 - <<PY
import csv, collections, glob
files = glob.glob("$OUT/prof_mvmem_$tag/**/run_counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(files[0]))) if files else []
agg = collections.defaultdict(list)
for r in rows:
    if "k_grid_mv_pieces" in r["Kernel_Name"] or "k_agg_mv_pieces" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, "launches", len(v), "first", v[0], "max", max(v))
if not agg: print("$C: no counters", open("$OUT/prof_mvmem_$tag.log").read()[-400:])
PY
done
find $ROOT/gpurun_out -name "*.csv" -size +20M -delete
