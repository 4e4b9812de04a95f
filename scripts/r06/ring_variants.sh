#!/bin/bash
# Split mode's ring of values (LEAN_GROUPS groups of four points a lane, LEAN_LOADS fetched per top-up): 8 / 6 (the product),
# 12 / 10, 16 / 12 - the mixed series at 1 % and the headline's fit (one lane per chunk, where the ring's LDS decides the waves per SIMD).
mkdir -p gpurun_out/r06
out=gpurun_out/r06/ring_variants.txt
: > $out
for v in product ring12 ring16; do
  lib=scripts/ab/${v}_libmdb_hip.so
  [ $v = product ] && lib=modelardb-rs_amd/csrc/libmdb_hip.so
  echo "== $v" >> $out
  MDB_HIP_LIBRARY=$PWD/$lib timeout 200 python3 scripts/r04/mixed_fit.py 1e9 rel1 2>&1 | grep "^rel1" | cut -c1-420 >> $out
  MDB_HIP_LIBRARY=$PWD/$lib timeout 300 python3 bench.py --timed fit --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --detail-file /tmp/d.json 2>/dev/null | python3 -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline fit', round(b['ms_per_step'],2), 'ms/step, kernel', round(b['roofline']['kernel_ms'],2))" >> $out 2>&1
done
cat $out
