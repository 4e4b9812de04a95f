#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python3 scripts/r04/fit_waves.py 16384 > $OUT/r04_fit_waves.log 2>&1
echo "fit_waves rc=$?"; cat $OUT/r04_fit_waves.log
PMC_TAG=lean timeout 1500 scripts/r04/pmc_kernel4.sh k_fit_models_lean scripts/profile_fit.py --series 1000 --points 10000000 > $OUT/r04_pmc_lean.log 2>&1
echo "pmc rc=$?"; cat $OUT/r04_pmc_lean.log
