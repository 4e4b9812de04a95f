#!/bin/bash
# Round 6's closing measurements, part A (one box): the bench lines the README names (compact line to *.json, everything
# measured to *_detail.json), then rocprofv3 statistics and the PMC passes of the default line (scripts/gpu_profile.sh)
# and the statistics of the fit line.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
run() { name=$1; shift; python3 bench.py --detail-file $OUT/${name}_detail.json "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? bytes=$(wc -c < $OUT/$name.json)"; }
run bench_default
run bench_timed_fit --timed fit
run bench_config4_shape_1gpu --series 12500 --points 1000000
run bench_config4_shape_timed_fit_1gpu --timed fit --series 12500 --points 1000000
run bench_config5_shape_1gpu --range-middle 0.5
bash scripts/gpu_profile.sh > $OUT/gpu_profile.log 2>&1; echo "gpu_profile rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_timed_fit -o timed_fit -- python3 $ROOT/bench.py --timed fit --steps 3 --warmup 1 --no-cpu-baseline --detail-file $OUT/prof_timed_fit_detail.json > $OUT/prof_timed_fit.log 2>&1
echo "timed fit trace rc=$?"
find $ROOT/gpurun_out -name "*.csv" -size +20M -delete
cat $OUT/bench_default.json; echo; cat $OUT/bench_timed_fit.json
