#!/bin/bash
# Round 5's closing measurements, part A (one box): the bench lines the README names, then rocprofv3 statistics and the
# PMC passes of the default line (scripts/gpu_profile.sh) and the statistics of the fit line.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd $ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default rc=$?"
python3 bench.py --timed fit > $OUT/bench_timed_fit.json 2> $OUT/bench_timed_fit.err; echo "timed fit rc=$?"
python3 bench.py --series 12500 --points 1000000 > $OUT/bench_config4_shape_1gpu.json 2> $OUT/bench_config4.err; echo "config 4 rc=$?"
python3 bench.py --timed fit --series 12500 --points 1000000 > $OUT/bench_config4_shape_timed_fit_1gpu.json 2> $OUT/bench_config4_fit.err; echo "config 4 fit rc=$?"
python3 bench.py --range-middle 0.5 > $OUT/bench_config5_shape_1gpu.json 2> $OUT/bench_config5.err; echo "config 5 rc=$?"
bash scripts/gpu_profile.sh > $OUT/gpu_profile.log 2>&1; echo "gpu_profile rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_timed_fit -o timed_fit -- python3 $ROOT/bench.py --timed fit --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_timed_fit.log 2>&1
echo "timed fit trace rc=$?"
find $ROOT/gpurun_out -name "*.csv" -size +20M -delete
tail -c 600 $OUT/bench_default.json; echo; tail -c 900 $OUT/bench_timed_fit.json
