#!/bin/bash
# rocprofv3 kernel statistics of the bench's mixed_models block (Constant / Linear / Random data, all three model
# types, lossless and 1 %: fit, grid and aggregates of 10^9 points) and of its host path; summaries to gpurun_out/.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mixed -o mixed -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-host-path --no-irregular > $OUT/prof_mixed.log 2>&1
echo "mixed rc=$?"
find $OUT -name "*.csv" -size +20M -delete
tail -c 300 $OUT/prof_mixed.log
