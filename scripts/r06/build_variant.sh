#!/bin/bash
# build_variant.sh NAME [-DFLAG ...]: a libmdb_hip.so whose mdb_fit.hip / mdb_grid.hip / mdb_agg.hip are compiled with the
# extra flags, written to scripts/ab/NAME_libmdb_hip.so (development tool; run in the container, the .so travels with
# gpurun; MDB_HIP_LIBRARY=<that file> selects it).
set -e
name=$1; shift
cd "$(dirname "$0")/../../modelardb-rs_amd/csrc"
make -s -j8 all
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -Wno-unused-function"
for unit in ${UNITS:-mdb_fit}; do
  /opt/rocm/bin/hipcc $flags "$@" -c $unit.hip -o /tmp/variant_${name}_$unit.o &
done
wait
objects=""
for unit in mdb_ctx mdb_grid mdb_agg mdb_fit mdb_synth mdb_comm; do
  if [ -f /tmp/variant_${name}_$unit.o ]; then objects="$objects /tmp/variant_${name}_$unit.o"; else objects="$objects $unit.o"; fi
done
mkdir -p ../../scripts/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/ab/${name}_libmdb_hip.so $objects mdb_pipeline.o mdb_mv_host_index.o -ldl -pthread -Wl,-rpath,/opt/rocm/lib
rm -f /tmp/variant_${name}_*.o
echo built scripts/ab/${name}_libmdb_hip.so
