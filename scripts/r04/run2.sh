#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; cd $ROOT
timeout 900 python3 bench.py --timed fit > $OUT/r04_bench_fit.json 2> $OUT/r04_bench_fit.err; echo "fit rc=$?"; tail -3 $OUT/r04_bench_fit.err; head -c 3000 $OUT/r04_bench_fit.json; echo
timeout 1200 python3 bench.py > $OUT/r04_bench_default.json 2> $OUT/r04_bench_default.err; echo "default rc=$?"; tail -3 $OUT/r04_bench_default.err; head -c 1500 $OUT/r04_bench_default.json; echo
