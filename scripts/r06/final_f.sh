#!/bin/bash
# Round 6, last pass: part B's measurements (the small calls, the mixed series kernel by kernel, segment files, one long
# chunk, the irregular series, SQ counters of the lossless wave kernel, the mixed block's kernel statistics) and the kernel
# statistics of the fit line, on the round's last sources.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
bash scripts/r06/final_b.sh 2>&1 | tail -40
timeout 300 env MDB_FIT_DEBUG=1 python3 scripts/r04/mixed_fit.py 1e9 rel1 2>&1 | grep -E "^\[fit\]" | sort -u | head -4 > $OUT/mixed_fit_rel1_counts.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_timed_fit -o timed_fit -- python3 $ROOT/bench.py --timed fit --steps 3 --warmup 1 --no-cpu-baseline --detail-file $OUT/prof_timed_fit_detail.json > $OUT/prof_timed_fit.log 2>&1
echo "timed fit trace rc=$?"
find $ROOT/gpurun_out -name "*.csv" -size +20M -delete
cat $OUT/mixed_fit_rel1_counts.log
