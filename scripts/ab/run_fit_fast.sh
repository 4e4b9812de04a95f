#!/bin/bash
# k_fit_models: plain / fast forms of the fitters / the lean kernel, at the real chunk size and with
# 4x the lanes (16 384-point chunks: what more lanes in flight could reach at best). Same box, alternating.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for chunk in 65536 16384; do
    for mode in plain fast lean; do
      case $mode in plain) F=0; L=0;; fast) F=1; L=0;; lean) F=1; L=1;; esac
      echo "== round $round chunk $chunk $mode"
      MDB_FIT_FAST=$F MDB_FIT_LEAN=$L MDB_FIT_PIECE_POINTS=1 python scripts/profile_fit.py --series 1000 --points 10000000 --chunk $chunk 2>&1 | grep -E "rep 1|k_fit_models" | tail -2
    done
  done
done
