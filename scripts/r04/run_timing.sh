#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  export MDB_HIP_LIBRARY=$PWD/scripts/ab/timing${v}_libmdb_hip.so
  python3 scripts/profile_fit.py --series ${SERIES:-1000} --points 10000000 2>&1 | grep -E "fit timing|k_fit_models " | tail -4
done
unset MDB_HIP_LIBRARY
python3 scripts/profile_fit.py --series ${SERIES:-1000} --points 10000000 2>&1 | grep -E "k_fit_models " | tail -1
