#!/bin/bash
# build_timing.sh: libmdb_hip.so variants with k_fit_models_lean's step regions timed (MDB_FIT_TIMING=0..5), written to
# scripts/ab/timingN_libmdb_hip.so (development tool, run in the container; the .so files travel with gpurun).
set -e
cd "$(dirname "$0")/../../modelardb-rs_amd/csrc"
make -s all
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -Wno-unused-function $EXTRA_FLAGS"
for n in "$@"; do
  /opt/rocm/bin/hipcc $flags -DMDB_FIT_TIMING=$n -c mdb_fit.hip -o /tmp/timing_fit_$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/ab/timing${n}_libmdb_hip.so mdb_ctx.o mdb_grid.o mdb_agg.o /tmp/timing_fit_$n.o mdb_synth.o mdb_comm.o mdb_pipeline.o mdb_mv_host_index.o -ldl -pthread -Wl,-rpath,/opt/rocm/lib
  echo built timing$n
done
