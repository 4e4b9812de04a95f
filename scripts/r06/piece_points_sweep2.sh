#!/bin/bash
mkdir -p gpurun_out/r06
out=gpurun_out/r06/piece_points_sweep2.txt
: > $out
run() { echo "== $*" >> $out; env "$@" timeout 200 python3 scripts/r04/mixed_fit.py 1e9 rel1 2>&1 | grep -E "^rel1" | cut -c1-500 | tail -1 >> $out; }
for pp in 256 512 768 1024 1536 2048 3072; do run MDB_FIT_PIECE_POINTS=$pp; done
cat $out
