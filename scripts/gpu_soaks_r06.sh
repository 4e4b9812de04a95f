# The randomised differential test (tests/test_gpu_soak.py: fit, grid, aggregates, ranges; host and resident batches,
# against the oracle) at length, in the modes that matter for round 6's code: the defaults (calls of a handful of chunks
# take the path without a device round trip), the same without that path, one wave per chunk everywhere with the host
# threads' cursors into every MacaqueV stream, pieces of 64 points and no cursors at all, and without the wave kernel,
# the range aggregates' pieces and anything a resident batch keeps; and (sixth) calls of 65 to 400 chunks with the groups of
# 64 chunks rotating in stretches of 1 to 512 steps and MacaqueV segments cut into blocks of 64 values.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
MDB_SOAK_CASES=4000 MDB_SOAK_HOST_CASES=200 timeout 1200 python -m pytest tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/r06/soak1.log 2>&1; grep -E "passed|failed" gpurun_out/r06/soak1.log | tail -1
MDB_FIT_SMALL=0 MDB_SOAK_CASES=3000 timeout 1200 python -m pytest tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/r06/soak2.log 2>&1; grep -E "passed|failed" gpurun_out/r06/soak2.log | tail -1
MDB_FIT_WAVE=1 MDB_GRID_MV_HOST_MIN_VALUES=1 MDB_SOAK_CASES=3000 timeout 1200 python -m pytest tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/r06/soak3.log 2>&1; grep -E "passed|failed" gpurun_out/r06/soak3.log | tail -1
MDB_FIT_WAVE=2 MDB_FIT_WAVE_WINDOW_POINTS=64 MDB_FIT_WAVE_POINTS_PER_STEP=40 MDB_FIT_PIECE_POINTS=64 MDB_GRID_MV_INDEX=0 MDB_SOAK_CASES=3000 timeout 1200 python -m pytest tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/r06/soak4.log 2>&1; grep -E "passed|failed" gpurun_out/r06/soak4.log | tail -1
MDB_FIT_WAVE=0 MDB_AGG_RANGE_PIECES=0 MDB_GRID_TS_CACHE=0 MDB_SOAK_CASES=2000 timeout 1200 python -m pytest tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/r06/soak5.log 2>&1; grep -E "passed|failed" gpurun_out/r06/soak5.log | tail -1
MDB_SOAK_ROTATING_CASES=400 MDB_SOAK_CASES=1 MDB_SOAK_HOST_CASES=1 timeout 1200 python -m pytest tests/test_gpu_soak.py -x -q -m gpu -k rotating > gpurun_out/r06/soak6.log 2>&1; grep -E "passed|failed" gpurun_out/r06/soak6.log | tail -1
# (seventh, round 6) the lossless wave path on every call of the soak that has a lossless bound: one wave per chunk, no small driver
MDB_FIT_WAVE=1 MDB_FIT_SMALL=0 MDB_SOAK_CASES=3000 timeout 1200 python -m pytest tests/test_gpu_soak.py -x -q -m gpu -k "random_series" > gpurun_out/r06/soak7.log 2>&1; grep -E "passed|failed" gpurun_out/r06/soak7.log | tail -1
# (eighth and ninth, round 6) split mode with k_fit_reject_flags' bits looked at however few are set: pieces of 64 and of 192 points
MDB_FIT_WAVE=0 MDB_FIT_SMALL=0 MDB_FIT_PIECE_POINTS=64 MDB_FIT_REJECT_FLAGS=1 MDB_SOAK_CASES=3000 timeout 1200 python -m pytest tests/test_gpu_soak.py -x -q -m gpu -k "random_series" > gpurun_out/r06/soak8.log 2>&1; grep -E "passed|failed" gpurun_out/r06/soak8.log | tail -1
MDB_FIT_WAVE=0 MDB_FIT_SMALL=0 MDB_FIT_PIECE_POINTS=192 MDB_FIT_REJECT_FLAGS=1 MDB_SOAK_CASES=3000 timeout 1200 python -m pytest tests/test_gpu_soak.py -x -q -m gpu -k "random_series" > gpurun_out/r06/soak9.log 2>&1; grep -E "passed|failed" gpurun_out/r06/soak9.log | tail -1
