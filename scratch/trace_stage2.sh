#!/bin/bash
# kernel trace of the Stage-2 step alone: trace_stage2.sh <tag>
R=$PWD
TAG=${1:-s2}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG} -o t --output-format csv -- python3 $R/bench.py --workload stage2 --no-extras --no-cpu-baseline --steps 5 --warmup 2 > $R/gpurun_out/${TAG}.log 2>&1 || { echo failed; tail -n 5 $R/gpurun_out/${TAG}.log; exit 1; }
tail -n 1 $R/gpurun_out/${TAG}.log | cut -c 1-300
