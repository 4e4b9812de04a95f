#!/bin/bash
# Round 6's closing measurements, part B (one box): SQ counters of the fit's model kernel, the small-call latencies, the
# mixed series kernel by kernel (fit, grid, aggregates), SQ counters of the lossless wave kernel and of k_agg_mv_chains,
# segment files end to end, one long lossless chunk.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
bash scripts/r05/pmc_fit.sh > $OUT/pmc_fit.log 2>&1; echo "pmc fit rc=$?"
cp $ROOT/gpurun_out/pmc_fit_models.json $ROOT/gpurun_out/pmc_fit_models.txt $OUT/ 2>/dev/null
timeout 600 python3 scripts/r04/fit_latency.py > $OUT/fit_latency.log 2>&1; echo "fit latency rc=$?"
timeout 600 python3 scripts/r04/mixed_fit.py 1e9 lossless,rel1 > $OUT/mixed_fit.log 2>&1; echo "mixed fit rc=$?"
MDB_FIT_DEBUG=1 timeout 600 python3 scripts/r04/mixed_fit.py 1e9 lossless 2>&1 | grep "k_fit_models_wave" | head -1 > $OUT/mixed_fit_wave_counts.log
timeout 600 python3 scripts/r04/mixed_grid.py > $OUT/mixed_grid.log 2>&1; echo "mixed grid rc=$?"
timeout 600 python3 scripts/r06/segment_files_e2e.py 64 $OUT/segment_files_e2e.json > $OUT/segment_files_e2e.log 2>&1; echo "segment files rc=$?"
timeout 600 python3 scripts/r06/lossless_long_chunk.py > $OUT/lossless_long_chunk.log 2>&1; echo "long chunk rc=$?"
timeout 600 python3 scripts/profile_irregular.py > $OUT/irregular.log 2>&1; echo "irregular rc=$?"
bash scripts/pmc_kernel.sh k_fit_models_wave scripts/r04/mixed_fit.py 1e9 lossless > $OUT/pmc_wave_lossless.log 2>&1; echo "pmc wave rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mixed -o mixed -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-irregular --no-cpu-baseline --detail-file $OUT/prof_mixed_detail.json > $OUT/prof_mixed.log 2>&1
echo "mixed trace rc=$?"
find $ROOT/gpurun_out -name "*.csv" -size +20M -delete
for f in fit_latency mixed_fit mixed_fit_wave_counts mixed_grid lossless_long_chunk pmc_wave_lossless; do echo "== $f"; tail -n 8 $OUT/$f.log | cut -c1-400; done
