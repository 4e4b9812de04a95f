#!/bin/bash
# Round 5's closing measurements, part C: the bench lines with the counters of part A / B in the tree (roofline.traffic,
# roofline.valu_issue), and the GPU tests.
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
python3 -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
