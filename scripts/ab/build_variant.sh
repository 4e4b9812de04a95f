#!/bin/bash
# build_variant.sh NAME [extra hipcc flags]: libmdb_hip.so with mdb_fit.hip compiled with the extra
# flags, written to scripts/ab/NAME_libmdb_hip.so (A/B tool, run in the container).
set -e
cd "$(dirname "$0")/../../modelardb-rs_amd/csrc"
name=$1; shift
make -s all
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero"
/opt/rocm/bin/hipcc $flags "$@" -c mdb_fit.hip -o /tmp/ab_fit_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/ab/${name}_libmdb_hip.so mdb_ctx.o mdb_grid.o mdb_agg.o /tmp/ab_fit_$name.o mdb_synth.o mdb_comm.o mdb_pipeline.o mdb_mv_host_index.o -ldl -pthread -Wl,-rpath,/opt/rocm/lib
echo built $name
