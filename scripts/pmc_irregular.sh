#!/bin/bash
# Hardware counters of the irregular-timestamp grid path (development tool; run via gpurun).
# One counter group per pass, each under its own timeout: a pass with FETCH_SIZE and WRITE_SIZE
# together once aborted and then sat there until gpurun killed it.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_irr_a -o a -- python3 $ROOT/scripts/profile_irregular.py > $OUT/pmc_irr_a.log 2>&1
echo "a rc=$?"
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_irr_b -o b -- python3 $ROOT/scripts/profile_irregular.py > $OUT/pmc_irr_b.log 2>&1
echo "b rc=$?"
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_irr_c -o c -- python3 $ROOT/scripts/profile_irregular.py > $OUT/pmc_irr_c.log 2>&1
echo "c rc=$?"
python3 - <<'PY'
import csv, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for name in ("a", "b", "c"):
    path = f"{out}/pmc_irr_{name}/{name}_counter_collection.csv"
    if not os.path.exists(path):
        print("missing", path); continue
    table = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0]
        if k in ("mdb::k_grid_prepass", "mdb::k_grid_serial", "mdb::k_grid_tiles"):
            table[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(table.items()):
        print(f"{k:24s} {c:22s} last={v[-1]:.4g} n={len(v)}")
PY
