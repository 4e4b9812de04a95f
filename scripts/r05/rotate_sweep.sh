#!/bin/bash
# Headline fit (bench.py --timed fit) under the rotation's switches, each line next to the plain kernel on the same box.
# usage: scripts/r05/rotate_sweep.sh OUT "ROTATE BLOCKS STEPS PAD" ...   (BLOCKS 0: the driver's own count)
out=$1; shift
mkdir -p "$(dirname "$out")"; : > "$out"
for cfg in "$@"; do
    set -- $cfg
    export MDB_FIT_ROTATE=$1 MDB_FIT_ROTATE_STEPS=$3 MDB_FIT_ROTATE_PAD=$4
    if [ "$2" != 0 ]; then export MDB_FIT_ROTATE_BLOCKS=$2; else unset MDB_FIT_ROTATE_BLOCKS; fi
    echo -n "rotate=$1 blocks=$2 steps=$3 pad=$4: " >> "$out"
    python bench.py --timed fit --no-host-path --steps 3 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d.get(\"ms_per_step\"),2), d[\"kernels_ms\"][\"k_fit_models_lean\"])" >> "$out" 2>&1
done
