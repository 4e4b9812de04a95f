#!/bin/bash
# Round 3's closing measurements on one box: the default bench line, rocprofv3 kernel statistics of the mixed-model
# block, of the fit of few chunks (k_fit_models_wave) and of the host path over mixed-model segments, the sweeps.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mixed -o mixed -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-irregular > $OUT/prof_mixed.log 2>&1
echo "mixed rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fit_sweep -o fit_sweep -- python3 $ROOT/scripts/profile_fit_sweep.py --bounds 50,10,1,0.5 --no-compare > $OUT/prof_fit_sweep.log 2>&1
echo "fit sweep rc=$?"
find $OUT -name "*.csv" -size +20M -delete
cd $ROOT
timeout 900 python3 scripts/profile_segment_lengths.py $OUT > $OUT/segment_lengths.log 2>&1
echo "segment lengths rc=$?"
