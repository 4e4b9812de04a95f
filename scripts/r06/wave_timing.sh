#!/bin/bash
# the wave kernel's cycles by region on the mixed series (timing build), then the product build's kernel times
cd ${GRAFT_REPO_ROOT:-/root/repo}
export MDB_HIP_LIBRARY=$PWD/scripts/ab/wavetiming_libmdb_hip.so
MDB_FIT_DEBUG=1 python3 scripts/r04/mixed_fit.py 1e9 lossless,rel1 2>&1 | grep -E "k_fit_models_wave|points," | awk '!seen[$0]++' | head -40
echo "--- MDB_FIT_WAVE=1 (never leave)"
MDB_FIT_WAVE=1 MDB_FIT_DEBUG=1 python3 scripts/r04/mixed_fit.py 1e9 rel1 2>&1 | grep -E "k_fit_models_wave|points," | awk '!seen[$0]++' | head -20
unset MDB_HIP_LIBRARY
echo "--- product"
python3 scripts/r04/mixed_fit.py 1e9 lossless,rel1 2>&1 | tail -3
