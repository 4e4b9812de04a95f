#!/bin/bash
# sample clocks and power while the fit runs
(for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done) > gpurun_out/clk_samples.txt &
SAMPLER=$!
python3 scripts/profile_fit.py --series 2048 --points 8388608 2>&1 | grep -E "rep|k_fit_models"
python3 scripts/profile_fit.py --series 512 --points 8388608 2>&1 | grep -E "rep|k_fit_models"
wait $SAMPLER
sort gpurun_out/clk_samples.txt | uniq -c | sort -rn | head -20
