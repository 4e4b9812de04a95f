#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "=== product"; python3 scripts/r04/mixed_grid.py 1e9 2>&1 | grep -E "grid:|timing|aggregates" | awk '!seen[$0]++' | head -12
for v in "$@"; do
  echo "=== $v"
  MDB_HIP_LIBRARY=$PWD/scripts/ab/${v}_libmdb_hip.so python3 scripts/r04/mixed_grid.py 1e9 2>&1 | grep -E "grid:|timing" | awk '!seen[$0]++' | head -12
done
