#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  echo "== $v"
  MDB_HIP_LIBRARY=$PWD/scripts/ab/${v}_libmdb_hip.so python scripts/profile_fit.py --series 1000 --points 10000000 2>&1 | grep -E "rep 1"
done
