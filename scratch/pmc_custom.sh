#!/bin/bash
# custom counter passes over the Stage-2 step: pmc_custom.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
R=$PWD
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $C -d $R/gpurun_out/pmc_${TAG}+stage2+c$i -o p --output-format csv -- python3 $R/bench.py --workload stage2 --no-extras --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_${TAG}+stage2+c$i.log 2>&1 || { echo "pass $i failed"; tail -n 5 $R/gpurun_out/pmc_${TAG}+stage2+c$i.log; exit 1; }
  echo "pass $i done"
done
cd $R
python3 scratch/pmc_summary.py gpurun_out "k_" ${TAG}+stage2+ > gpurun_out/${TAG}_pmc_custom.txt
grep -E "k_critic_bwd_mfma|k_gp_" gpurun_out/${TAG}_pmc_custom.txt
