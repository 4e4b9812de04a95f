# (scripts/ab/libmdb_hip_old.so: `git show <commit before>:modelardb-rs_amd/csrc/mdb_grid.hip` into a copy of csrc/, make,
# copy the library here; it is not kept in the tree)
# A/B on one box: k_grid_fused with one atomic per wave for the list of segments with serial work (the library built
# from the commit before, scripts/ab/libmdb_hip_old.so) against the list gathered per round in LDS / no list at all.
for round in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export MDB_HIP_LIBRARY=$PWD/scripts/ab/libmdb_hip_old.so; else unset MDB_HIP_LIBRARY; fi
  mkdir -p gpurun_out/ab_$lib
  timeout 600 python3 scripts/profile_segment_lengths.py gpurun_out/ab_$lib lengths > gpurun_out/ab_$lib/log.txt 2>&1
  timeout 600 python3 scripts/profile_segment_lengths.py gpurun_out/ab_$lib sweep >> gpurun_out/ab_$lib/log.txt 2>&1
  echo "$lib round $round"; cut -d, -f1,8 gpurun_out/ab_$lib/segment_lengths.csv | tr '\n' ' '; echo; cut -d, -f1,7 gpurun_out/ab_$lib/error_bound_sweep.csv | tail -5 | tr '\n' ' '; echo
done
done
