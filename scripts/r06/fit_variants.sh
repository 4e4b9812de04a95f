#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "=== product"; python3 scripts/r04/mixed_fit.py 1e9 ${BOUNDS:-lossless} 2>&1 | tail -1
for v in "$@"; do
  echo "=== $v"
  MDB_HIP_LIBRARY=$PWD/scripts/ab/${v}_libmdb_hip.so python3 scripts/r04/mixed_fit.py 1e9 ${BOUNDS:-lossless} 2>&1 | tail -1
done
