#!/bin/bash
# A/B two builds of libmdb_hip.so on the same box, interleaved (rule: never compare across boxes).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2 3; do
  for v in old new; do
    echo "== $v round $round"
    MDB_HIP_LIBRARY=$PWD/scripts/ab/${v}_libmdb_hip.so python scripts/profile_grid.py --distinct 8 --points 2000000 --tile 128 --steps 5 2>&1 | grep -E "k_grid_tiles"
  done
done
