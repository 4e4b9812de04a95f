#!/bin/bash
# The mixed series at 1 %: when a wave leaves its chunk to split mode (MDB_FIT_WAVE_POINTS_PER_STEP, the window it looks
# at), the wave kernel alone, split mode alone.
mkdir -p gpurun_out/r06
out=gpurun_out/r06/mixed_fit_leave_sweep.txt
: > $out
run() { echo "== $*" >> $out; env "$@" MDB_FIT_DEBUG=1 timeout 200 python3 scripts/r04/mixed_fit.py 1e9 rel1 2>&1 | grep -E "^rel1|k_fit_models_wave:" | cut -c1-700 | tail -2 >> $out; }
run MDB_X=0
for pps in 5 10 40 80; do run MDB_FIT_WAVE_POINTS_PER_STEP=$pps; done
for window in 256 4096; do run MDB_FIT_WAVE_WINDOW_POINTS=$window; done
run MDB_FIT_WAVE=1
run MDB_FIT_WAVE=0
cat $out
