#!/bin/bash
# Evidence for the default bench line (run from the repo root on the GPU box): collect_r2.sh <tag>
# kernel-trace statistics of the default command (Stage-2 headline + the extra workloads), then counter passes
# (one run each) over the Stage-2 step for HBM bytes and instruction mix.
R=$PWD
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_all -o all --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_all.log 2>&1 || { echo "kernel trace failed"; tail -n 5 $R/gpurun_out/${TAG}_all.log; exit 1; }
for P in fetch write sq; do
  case $P in
    sq) C="SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES";;
    fetch) C="FETCH_SIZE";;
    write) C="WRITE_SIZE";;
  esac
  timeout -s KILL 150 rocprofv3 --pmc $C -d $R/gpurun_out/pmc_${TAG}_$P -o p --output-format csv -- python3 $R/bench.py --workload stage2 --no-extras --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_${TAG}_$P.log 2>&1 || { echo "pass $P failed"; tail -n 5 $R/gpurun_out/pmc_${TAG}_$P.log; exit 1; }
done
cd $R
python3 scratch/pmc_summary.py gpurun_out "k_" ${TAG}_ > gpurun_out/${TAG}_pmc_summary.txt
cat gpurun_out/${TAG}_pmc_summary.txt
