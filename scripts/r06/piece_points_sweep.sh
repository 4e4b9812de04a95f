#!/bin/bash
# The mixed series at 1 %, split mode alone: points per piece (the default asks for two waves a SIMD), with and without k_fit_reject_flags.
mkdir -p gpurun_out/r06
out=gpurun_out/r06/piece_points_sweep.txt
: > $out
run() { echo "== $*" >> $out; env "$@" timeout 200 python3 scripts/r04/mixed_fit.py 1e9 rel1 2>&1 | grep -E "^rel1" | cut -c1-700 | tail -1 >> $out; }
for pp in 512 1024 2048 4096; do
  run MDB_FIT_PIECE_POINTS=$pp
  run MDB_FIT_PIECE_POINTS=$pp MDB_FIT_REJECT_FLAGS=0
done
cat $out
