#!/bin/bash
# End-state evidence: kernel stats for the default bench line and the spectral workload, PMC passes over the
# fused Stage-1 kernel, the bench line with all workloads.  Run from the repo root on the GPU box.
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/fin_s1 -o s1 --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/fin_s1.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/fin_sp -o sp --output-format csv -- python3 $R/bench.py --workload spectral --no-cpu-baseline > $R/gpurun_out/fin_sp.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/fin_s2 -o s2 --output-format csv -- python3 $R/bench.py --workload stage2 --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/fin_s2.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/fin_pmc_fetch -o p --output-format csv -- python3 $R/scratch/run_bwd.py 3 > $R/gpurun_out/fin_pmc1.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/fin_pmc_write -o p --output-format csv -- python3 $R/scratch/run_bwd.py 3 > $R/gpurun_out/fin_pmc2.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/fin_pmc_sq -o p --output-format csv -- python3 $R/scratch/run_bwd.py 3 > $R/gpurun_out/fin_pmc3.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $R/gpurun_out/fin_pmc_grbm -o p --output-format csv -- python3 $R/scratch/run_bwd.py 3 > $R/gpurun_out/fin_pmc4.log 2>&1
cd $R
timeout -k 10 400 python bench.py --extras > gpurun_out/fin_extras.json 2> gpurun_out/fin_extras.err
echo collected
