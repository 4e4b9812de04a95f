#!/bin/bash
# Dev tool (runs on the GPU box): rebuild mdb_grid.o with different output tile sizes and time the
# grid kernels on the same oracle-fitted workload.
cd ${GRAFT_REPO_ROOT:-/root/repo}/modelardb-rs_amd/csrc
for T in 4096 8192 16384 32768; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -DMDB_TILE_POINTS=$T -c mdb_grid.hip -o mdb_grid.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmdb_hip.so mdb_ctx.o mdb_grid.o mdb_agg.o mdb_fit.o mdb_synth.o -Wl,-rpath,/opt/rocm/lib
  echo "== TILE_POINTS=$T"
  (cd ../.. && python scripts/profile_grid.py --distinct 8 --points 2000000 --tile 128 --steps 5 2>&1 | grep -E "k_grid_tiles|wall per step")
done
