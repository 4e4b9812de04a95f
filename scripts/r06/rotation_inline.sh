#!/bin/bash
# The rotation queue's take / give INLINED into k_fit_models_lean, with and without the compiler barriers: the tests of
# tests/test_gpu_fit_rotation.py and the headline fit's time under each build. (Build the variants in the container first:
#   scripts/r06/build_variant.sh rot_inline -DMDB_ROTATION_INLINE
#   scripts/r06/build_variant.sh rot_inline_nobarrier -DMDB_ROTATION_INLINE -DMDB_ROTATION_NO_BARRIER )
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in ${VARIANTS:-rot_none rot_w_take_compiler_only}; do
  echo "=== $v"
  export MDB_HIP_LIBRARY=$PWD/scripts/ab/${v}_libmdb_hip.so
  timeout 300 python3 -m pytest tests/test_gpu_fit_rotation.py -x -q 2>&1 | tail -4
  timeout 300 python3 scripts/profile_fit.py --series 1000 --points 10000000 2>&1 | grep -E "k_fit_models|fit:" | tail -3
done
unset MDB_HIP_LIBRARY
echo "=== product"
python3 scripts/profile_fit.py --series 1000 --points 10000000 2>&1 | grep -E "k_fit_models|fit:" | tail -3
