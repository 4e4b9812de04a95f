#!/bin/bash
mkdir -p gpurun_out/r06
out=gpurun_out/r06/probe_try.txt
: > $out
run() { echo "== $*" >> $out; env "$@" timeout 200 python3 scripts/r04/mixed_fit.py 1e9 rel1 2>&1 | grep -E "^rel1|probe:|reject_flags:" | cut -c1-700 | sort -u | tail -3 >> $out; }
run MDB_FIT_DEBUG=1
run MDB_FIT_WAVE_PROBE=0
run MDB_FIT_REJECT_FLAGS=0
run MDB_FIT_WAVE_PROBE=0 MDB_FIT_REJECT_FLAGS=0
echo "== sine + noise, defaults" >> $out
timeout 400 python3 scripts/profile_fit_sweep.py --bounds 2,0.7,0.5,0.3,0.1 2>&1 | cut -c1-420 >> $out
cat $out
