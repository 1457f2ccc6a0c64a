#!/bin/bash
# Where the waves of a workload spend their cycles: pmc_wait.sh <workload> <tag>
R=$PWD
W=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_${TAG}+wait -o p --output-format csv -- python3 $R/bench.py --workload $W --no-extras --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_${TAG}+wait.log 2>&1 || { echo "pass failed"; tail -n 5 $R/gpurun_out/pmc_${TAG}+wait.log; exit 1; }
cd $R
python3 scratch/pmc_summary.py gpurun_out "k_" ${TAG}+wait
