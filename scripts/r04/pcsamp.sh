#!/bin/bash
# PC sampling of the fit kernel (beta feature of rocprofv3): which instructions of k_fit_models_lean the waves sit on.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
METHOD=${1:-host_trap}; UNIT=${2:-time}; INTERVAL=${3:-100}
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pcsamp_$METHOD
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INTERVAL \
   --output-format csv -d $OUT/pcsamp_$METHOD -o run -- python3 $ROOT/scripts/profile_fit.py --series 200 --points 10000000 > $OUT/pcsamp_$METHOD.log 2>&1
echo "rc=$?"
tail -5 $OUT/pcsamp_$METHOD.log
ls -la $OUT/pcsamp_$METHOD/* | head
for f in $OUT/pcsamp_$METHOD/*/*pc_sampling*.csv $OUT/pcsamp_$METHOD/*pc_sampling*.csv; do [ -f "$f" ] && head -5 "$f" && wc -l "$f"; done
find $OUT/pcsamp_$METHOD -size +30M -delete
