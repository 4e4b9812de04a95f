# The fit of the mixed series (bench.py's mixed_models block) with MacaqueV gaps going to the wave encoder from
# 256 (default) / 128 / 64 / 32 values on.
for g in 256 128 64 32; do
  MDB_FIT_GAP_MIN_VALUES=$g timeout 400 python bench.py --no-irregular --no-host-path --steps 2 > gpurun_out/bench_gap$g.json 2>/dev/null
  echo "gap min values $g"; python3 scripts/show_mixed_fit.py gpurun_out/bench_gap$g.json | head -2
done
