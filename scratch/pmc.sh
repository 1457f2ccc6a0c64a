#!/bin/bash
# Counter passes over one bench workload (run from the repo root on the GPU box): pmc.sh <workload> <tag> [pass...]
# passes: sq (instruction mix), fetch, write (HBM bytes).  Each pass is a run of its own.
R=$PWD
W=$1; TAG=$2; shift 2
PASSES=${@:-sq}
cd /tmp && export TMPDIR=/tmp
for P in $PASSES; do
  case $P in
    sq) C="SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES";;
    fetch) C="FETCH_SIZE";;
    write) C="WRITE_SIZE";;
    *) echo "unknown pass $P"; exit 1;;
  esac
  timeout -s KILL 120 rocprofv3 --pmc $C -d $R/gpurun_out/pmc_${TAG}_$P -o p --output-format csv -- python3 $R/bench.py --workload $W --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/pmc_${TAG}_$P.log 2>&1 || { echo "pass $P failed"; tail -n 5 $R/gpurun_out/pmc_${TAG}_$P.log; exit 1; }
done
cd $R
python3 scratch/pmc_summary.py gpurun_out "k_" $TAG
