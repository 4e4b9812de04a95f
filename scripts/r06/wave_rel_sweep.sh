#!/bin/bash
# mixed series at 1 %: how eagerly k_fit_models_wave leaves chunks to split mode (MDB_FIT_WAVE_POINTS_PER_STEP)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for p in 20 14 10 6; do
  echo "points_per_step $p"; MDB_FIT_WAVE_POINTS_PER_STEP=$p python3 scripts/r04/mixed_fit.py 1e9 rel1 2>&1 | tail -1
done
echo "sine, few chunks (fit latency rows)"; python3 scripts/r04/fit_latency.py 2>&1 | tail -12
