#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_spectral_quant.py tests/test_gpu_full_size.py tests/test_gpu_edge_cases.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 100 python scratch/tk.py q88 reps=20 2>/dev/null | grep q88
