#!/bin/bash
# split mode (speculative pieces) of the headline fit with smaller LDS rings, so that all waves are resident
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in default ring6 ring5 ring4; do
  if [ $v = default ]; then unset MDB_HIP_LIBRARY; else export MDB_HIP_LIBRARY=$PWD/scripts/ab/${v}_libmdb_hip.so; fi
  for pp in 1 32768 24576 16384; do
    printf "== %-8s piece_points %-6s " $v $pp
    MDB_FIT_WAVE=0 MDB_FIT_PIECE_POINTS=$pp python3 scripts/profile_fit.py --series 1000 --points 10000000 2>&1 | grep -E "rep 1|k_fit_models|k_fit_walk" | tr '\n' ' '; echo
  done
done
