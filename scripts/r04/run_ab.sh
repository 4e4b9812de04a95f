#!/bin/bash
# run_ab.sh NAME...: k_fit_models* of the headline fit under scripts/ab/NAME_libmdb_hip.so ("default" = the tree's), twice
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for v in "$@"; do
    if [ $v = default ]; then unset MDB_HIP_LIBRARY; else export MDB_HIP_LIBRARY=$PWD/scripts/ab/${v}_libmdb_hip.so; fi
    printf "== round $round %-10s" $v
    python3 scripts/profile_fit.py --series ${SERIES:-1000} --points 10000000 2>&1 | grep -E "k_fit_models" | tail -1
  done
done
