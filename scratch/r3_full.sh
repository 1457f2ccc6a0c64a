#!/bin/bash
set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_gpu_tests.log 2>&1 || { tail -n 40 gpurun_out/r3_gpu_tests.log; exit 1; }
tail -n 2 gpurun_out/r3_gpu_tests.log
timeout -k 10 120 python -u scratch/tk.py gen reps=20 > gpurun_out/dig_bwd_tk.log 2>&1 || { tail -n 20 gpurun_out/dig_bwd_tk.log; exit 1; }
cat gpurun_out/dig_bwd_tk.log
