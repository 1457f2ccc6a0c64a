#!/bin/bash
set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_generator.py tests/test_gpu_chain.py tests/test_gpu_steps.py tests/test_gpu_full_size.py -x -q --timeout 120 --timeout-method thread > gpurun_out/dig_bwd_tests.log 2>&1 || { tail -n 40 gpurun_out/dig_bwd_tests.log; exit 1; }
tail -n 2 gpurun_out/dig_bwd_tests.log
timeout -k 10 120 python -u scratch/tk.py gen reps=20 > gpurun_out/dig_bwd_tk.log 2>&1 || { tail -n 20 gpurun_out/dig_bwd_tk.log; exit 1; }
cat gpurun_out/dig_bwd_tk.log
