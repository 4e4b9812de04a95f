#!/bin/bash
mkdir -p gpurun_out/r06
out=gpurun_out/r06/host_depth_walls.txt
: > $out
for setting in "2 1" "3 2" "3 3" "4 3" "4 4" "2 1" "3 2" "4 3"; do
  set -- $setting
  MDB_GRID_PIPELINE_CONTEXTS=$1 MDB_HOST_GRID_PREFETCH=$2 timeout 300 python3 scripts/r06/host_depth_walls.py 2>&1 | grep -v "^\[" | tail -4 >> $out
done
cat $out
