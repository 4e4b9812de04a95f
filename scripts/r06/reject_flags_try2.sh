#!/bin/bash
mkdir -p gpurun_out/r06
out=gpurun_out/r06/reject_flags_try2.txt
: > $out
run() { echo "== $*" >> $out; env "$@" timeout 200 python3 scripts/r04/mixed_fit.py 1e9 rel1 2>&1 | grep -E "^rel1" | cut -c1-700 | tail -1 >> $out; }
for w in 2 8 16 32; do
  run MDB_FIT_SPLIT_WAVES_PER_SIMD=$w
  run MDB_FIT_SPLIT_WAVES_PER_SIMD=$w MDB_FIT_REJECT_FLAGS=0
  run MDB_FIT_SPLIT_WAVES_PER_SIMD=$w MDB_FIT_WAVE=0
  run MDB_FIT_SPLIT_WAVES_PER_SIMD=$w MDB_FIT_WAVE=0 MDB_FIT_REJECT_FLAGS=0
done
cat $out
