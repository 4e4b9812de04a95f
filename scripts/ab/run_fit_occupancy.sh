#!/bin/bash
# How does k_fit_models scale with waves per SIMD? Same points, smaller chunks = more lanes.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for chunk in 65536 32768 16384 8192; do
  echo "== chunk $chunk"
  MDB_FIT_PIECE_POINTS=1 python scripts/profile_fit.py --series 1000 --points 10000000 --chunk $chunk 2>&1 | grep -E "rep 1|k_fit_models"  | tail -2
done
