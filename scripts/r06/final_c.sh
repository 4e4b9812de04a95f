#!/bin/bash
# Round 6's closing measurements, part C: the bench lines with the counters of part A / B in the tree (roofline.traffic,
# the fit's valu_issue), the GPU tests and smoke().
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
run bench_driver_args --gpus 1 --steps 20 --warmup 5
python3 -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cat $OUT/bench_default.json
