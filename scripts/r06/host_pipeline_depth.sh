#!/bin/bash
# The mixed series as host batches through GridStream: contexts of the library's pipeline x submits ahead of the stream.
mkdir -p gpurun_out/r06
out=gpurun_out/r06/host_pipeline_depth.txt
: > $out
for contexts in 2 3 4; do
  for ahead in 1 2 3; do
    echo "== MDB_GRID_PIPELINE_CONTEXTS=$contexts MDB_HOST_GRID_PREFETCH=$ahead" >> $out
    MDB_GRID_PIPELINE_CONTEXTS=$contexts MDB_HOST_GRID_PREFETCH=$ahead timeout 200 python3 scripts/r06/host_mixed_phases.py 2>&1 | cut -c1-900 >> $out
  done
done
cat $out
