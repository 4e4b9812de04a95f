#!/bin/bash
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_fit_reject_flags.py tests/test_gpu_fit_small.py tests/test_gpu_fit_long.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python3 scripts/r04/mixed_fit.py 1e9 rel1 2>&1 | grep "^rel1" | cut -c1-400
MDB_FIT_DEBUG=1 timeout 600 python3 scripts/r06/probe_rough_smooth.py 2>&1 | grep -E "^rough" | cut -c1-1300 > gpurun_out/r06/probe_rough_smooth.txt; cut -c1-700 gpurun_out/r06/probe_rough_smooth.txt
timeout 400 python3 scripts/profile_fit_sweep.py --bounds 2,0.7,0.5,0.3,0.1 2>&1 | cut -c1-330 > gpurun_out/r06/probe_sine_sweep.txt; cat gpurun_out/r06/probe_sine_sweep.txt
