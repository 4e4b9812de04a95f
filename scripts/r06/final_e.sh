#!/bin/bash
# Round 6, after the aggregates' MacaqueV path was rebuilt: the mixed series kernel by kernel again (fit, grid, aggregates)
# and the rocprofv3 statistics of a bench run with the mixed block in it.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
timeout 300 python3 scripts/r04/mixed_fit.py 1e9 lossless,rel1 > $OUT/mixed_fit.log 2>&1; echo "mixed fit rc=$?"
timeout 300 python3 scripts/r04/mixed_grid.py > $OUT/mixed_grid.log 2>&1; echo "mixed grid rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mixed -o mixed -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-irregular --no-cpu-baseline --detail-file $OUT/prof_mixed_detail.json > $OUT/prof_mixed.log 2>&1
echo "mixed trace rc=$?"
find $ROOT/gpurun_out -name "*.csv" -size +20M -delete
tail -n 6 $OUT/mixed_grid.log | cut -c1-300
