cd $GRAFT_REPO_ROOT
MDB_SOAK_CASES=4000 MDB_SOAK_HOST_CASES=200 timeout 900 python -m pytest tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/soak1.log 2>&1; grep -E "passed|failed" gpurun_out/soak1.log | tail -1
MDB_FIT_WAVE=1 MDB_GRID_MV_HOST_MIN_VALUES=1 MDB_SOAK_CASES=3000 timeout 900 python -m pytest tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/soak2.log 2>&1; grep -E "passed|failed" gpurun_out/soak2.log | tail -1
MDB_FIT_WAVE=2 MDB_FIT_WAVE_WINDOW_POINTS=64 MDB_FIT_WAVE_POINTS_PER_STEP=40 MDB_FIT_PIECE_POINTS=64 MDB_GRID_MV_INDEX=0 MDB_SOAK_CASES=3000 timeout 900 python -m pytest tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/soak3.log 2>&1; grep -E "passed|failed" gpurun_out/soak3.log | tail -1
MDB_FIT_WAVE=0 MDB_AGG_RANGE_PIECES=0 MDB_GRID_TS_CACHE=0 MDB_SOAK_CASES=2000 timeout 900 python -m pytest tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/soak4.log 2>&1; grep -E "passed|failed" gpurun_out/soak4.log | tail -1
