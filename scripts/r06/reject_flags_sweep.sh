#!/bin/bash
# Sine + noise (4 096 chunks of 65 536 points) under relative bounds from far above the noise to far below it: the
# library's choice, the wave kernel alone, split mode alone - without k_fit_reject_flags, with it, with it and smaller pieces.
mkdir -p gpurun_out/r06
out=gpurun_out/r06/reject_flags_sweep.txt
: > $out
for setting in "MDB_FIT_REJECT_FLAGS=0" "MDB_X=1" "MDB_FIT_SPLIT_WAVES_PER_SIMD=16"; do
  echo "== $setting" >> $out
  env $setting timeout 400 python3 scripts/profile_fit_sweep.py --bounds 10,2,1,0.7,0.5,0.3,0.1 2>&1 | cut -c1-600 >> $out
done
cat $out
