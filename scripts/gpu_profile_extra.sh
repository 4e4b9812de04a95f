#!/bin/bash
# rocprofv3 kernel statistics of the fit kernels (benchmark workload) and of the lossless / MacaqueV
# path (BASELINE config 1); summaries go to gpurun_out/, the ones worth keeping are copied to profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fit -o fit -- python3 $ROOT/scripts/profile_fit.py --series 1000 --points 10000000 > $OUT/prof_fit.log 2>&1
echo "fit rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_lossless -o lossless -- python3 $ROOT/scripts/profile_grid.py --error-bound 0 --distinct 1 --points 1000000 --tile 1 --steps 5 > $OUT/prof_lossless.log 2>&1
echo "lossless grid rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_lossless_fit -o lossless_fit -- python3 $ROOT/scripts/profile_lossless_fit.py > $OUT/prof_lossless_fit.log 2>&1
echo "lossless fit rc=$?"
find $OUT -name "*.csv" -size +20M -delete
for f in fit lossless lossless_fit; do tail -n 2 $OUT/prof_$f.log; done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_irregular -o irregular -- python3 $ROOT/scripts/profile_irregular.py > $OUT/prof_irregular.log 2>&1
echo "irregular rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_lossless_agg -o lossless_agg -- python3 $ROOT/scripts/profile_lossless_agg.py > $OUT/prof_lossless_agg.log 2>&1
echo "lossless agg rc=$?"
find $OUT -name "*.csv" -size +20M -delete
for f in irregular lossless_agg; do tail -n 3 $OUT/prof_$f.log; done
