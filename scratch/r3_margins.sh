#!/bin/bash
# how much of each tolerance the GPU parity tests use (DPD_CLOSE_REPORT), whole GPU suite
mkdir -p gpurun_out
DPD_CLOSE_REPORT=1 DPD_CLOSEQ_REPORT=1 timeout -k 10 560 python -u -m pytest tests -m gpu -x -q -s --timeout 120 --timeout-method thread > gpurun_out/r3_margins.log 2>&1
rc=$?
echo "exit $rc"; tail -3 gpurun_out/r3_margins.log
exit $rc
