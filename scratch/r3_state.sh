#!/bin/bash
# GPU suite, then the default bench line and its kernel-trace statistics
mkdir -p gpurun_out
timeout -k 10 560 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_state_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r3_state_tests.log
[ $rc -eq 0 ] || exit $rc
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3_state_all -o all --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r3_state_bench.json 2> $R/gpurun_out/r3_state_bench.err || { tail -5 $R/gpurun_out/r3_state_bench.err; exit 1; }
cd $R
cat gpurun_out/r3_state_bench.json | head -c 3000
