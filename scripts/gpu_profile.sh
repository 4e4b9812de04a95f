#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py, then two PMC passes
# (FETCH_SIZE and WRITE_SIZE need separate passes on gfx950: TCC has 4 slots, they cost 3 + 2).
# Results land in gpurun_out/prof_*; the summaries worth keeping are copied to profiles/ by hand.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="${BENCH_ARGS:---steps 3 --warmup 1 --no-cpu-baseline}"
# The sources the counters below are measured on: bench.py reports roofline.traffic only while they are
# the ones in the tree (scripts/summarize_profiles.py copies the hash into profiles/pmc_grid_tiles.json).
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.source_hash(bench.GRID_KERNEL_SOURCES))" > $OUT/prof_source_hash.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o trace -- python3 $ROOT/bench.py $BENCH_ARGS > $OUT/prof_trace.log 2>&1
echo "trace rc=$?"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -o fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/prof_fetch.log 2>&1
echo "fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -o write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/prof_write.log 2>&1
echo "write rc=$?"
ls -R $OUT | head -50
# keep only the small summaries (the per-dispatch trace of 3 steps is small too)
find $OUT -name "*.csv" -size +20M -delete
tail -3 $OUT/prof_trace.log
