#!/bin/bash
# GPU suite (with the tolerance-use report) and the Q8.8 bench line
mkdir -p gpurun_out
DPD_CLOSE_REPORT=1 timeout -k 10 560 python -u -m pytest tests -m gpu -x -q -s --timeout 120 --timeout-method thread > gpurun_out/r3_margins2.log 2>&1
rc=$?
tail -3 gpurun_out/r3_margins2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --workload q88 --no-cpu-baseline > gpurun_out/r3_q88_line.json 2> gpurun_out/r3_q88_line.err || { tail -5 gpurun_out/r3_q88_line.err; exit 1; }
cat gpurun_out/r3_q88_line.json
