#!/bin/bash
# full GPU suite, then the evidence for the default bench line (see collect_r2.sh)
set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_gpu_tests.log 2>&1 || { tail -n 40 gpurun_out/r3_gpu_tests.log; exit 1; }
tail -n 2 gpurun_out/r3_gpu_tests.log
bash scratch/collect_r2.sh r3 extras || exit 1
timeout -k 10 400 python -u bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err || { tail -n 20 gpurun_out/r3_bench.err; exit 1; }
tail -c 600 gpurun_out/r3_bench.json
