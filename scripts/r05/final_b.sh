#!/bin/bash
# Round 5's closing measurements, part B (one box): SQ counters of the fit's model kernel, the sweeps, the small-call
# latencies, the mixed series kernel by kernel, segment files end to end, irregular timestamps.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd $ROOT
bash scripts/r05/pmc_fit.sh > $OUT/pmc_fit.log 2>&1; echo "pmc fit rc=$?"
cp $ROOT/gpurun_out/pmc_fit_models.json $ROOT/gpurun_out/pmc_fit_models.txt $OUT/ 2>/dev/null
timeout 1200 python3 scripts/profile_segment_lengths.py $OUT > $OUT/segment_lengths.log 2>&1; echo "segment lengths rc=$?"
timeout 900 python3 scripts/profile_fit_sweep.py --out $OUT/fit_few_chunks.csv > $OUT/fit_few_chunks.log 2>&1; echo "fit sweep rc=$?"
timeout 600 python3 scripts/r04/fit_latency.py > $OUT/fit_latency.log 2>&1; echo "fit latency rc=$?"
timeout 600 python3 scripts/r05/fit_crossover.py > $OUT/fit_crossover.log 2>&1; echo "fit crossover rc=$?"
timeout 600 python3 scripts/r04/mixed_fit.py 1e9 lossless,rel1 > $OUT/mixed_fit.log 2>&1; echo "mixed fit rc=$?"
timeout 600 python3 scripts/r04/mixed_grid.py > $OUT/mixed_grid.log 2>&1; echo "mixed grid rc=$?"
timeout 600 python3 scripts/r04/segment_files_e2e.py > $OUT/segment_files_e2e.log 2>&1; echo "segment files rc=$?"
timeout 600 python3 scripts/profile_irregular.py > $OUT/irregular.log 2>&1; echo "irregular rc=$?"
timeout 600 python3 scripts/r04/tail_lengths.py > $OUT/tail_lengths.log 2>&1; echo "tail lengths rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mixed -o mixed -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-irregular --no-cpu-baseline > $OUT/prof_mixed.log 2>&1
echo "mixed trace rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fit_latency -o fit_latency -- python3 $ROOT/scripts/r04/fit_latency.py > $OUT/prof_fit_latency.log 2>&1
echo "fit latency trace rc=$?"
find $ROOT/gpurun_out -name "*.csv" -size +20M -delete
for f in fit_latency fit_crossover mixed_fit mixed_grid segment_files_e2e irregular tail_lengths; do echo "== $f"; tail -n 8 $OUT/$f.log | cut -c1-300; done
