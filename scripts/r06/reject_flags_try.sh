#!/bin/bash
# The mixed series at 1 % with k_fit_reject_flags: split mode alone, the wave kernel leaving early, the default; and without.
mkdir -p gpurun_out/r06
out=gpurun_out/r06/reject_flags_try.txt
: > $out
run() { echo "== $*" >> $out; env "$@" timeout 200 python3 scripts/r04/mixed_fit.py 1e9 rel1 2>&1 | grep -E "^rel1" | cut -c1-700 | tail -1 >> $out; }
run MDB_X=0
run MDB_FIT_WAVE=0
run MDB_FIT_WAVE_POINTS_PER_STEP=80
run MDB_FIT_WAVE_POINTS_PER_STEP=40
run MDB_FIT_REJECT_FLAGS=0
run MDB_FIT_REJECT_FLAGS=0 MDB_FIT_WAVE=0
cat $out
