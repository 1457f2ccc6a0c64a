#!/bin/bash
# Evidence for the default bench line (run from the repo root on the GPU box): collect_r2.sh <tag> [extras]
# 1. kernel-trace statistics of the default command (Stage-2 headline + the extra workloads);
# 2. counter passes (one run each: FETCH_SIZE, WRITE_SIZE, SQ instruction mix) over the Stage-2 step;
# 3. with "extras": FETCH_SIZE / WRITE_SIZE passes over each extra workload.
R=$PWD
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_all -o all --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_all.log 2>&1 || { echo "kernel trace failed"; tail -n 5 $R/gpurun_out/${TAG}_all.log; exit 1; }
echo "kernel trace done"
pass() {   # pass <workload> <name> <counters...>
  local W=$1 N=$2; shift 2
  timeout -s KILL 150 rocprofv3 --pmc "$@" -d $R/gpurun_out/pmc_${TAG}+${W}+$N -o p --output-format csv -- python3 $R/bench.py --workload $W --no-extras --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_${TAG}+${W}+$N.log 2>&1 || { echo "pass $W $N failed"; tail -n 5 $R/gpurun_out/pmc_${TAG}+${W}+$N.log; exit 1; }
  echo "pass $W $N done"
}
pass stage2 fetch FETCH_SIZE || exit 1
pass stage2 write WRITE_SIZE || exit 1
pass stage2 sq SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES || exit 1
pass stage2 grbm GRBM_GUI_ACTIVE || exit 1
if [ "$2" = "extras" ]; then
  for W in stage1 stage1_qat q88 gen_pa gen_pa_float gen_pa_train spectral; do
    pass $W fetch FETCH_SIZE || exit 1
    pass $W write WRITE_SIZE || exit 1
    pass $W sq SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES || exit 1
  done
fi
cd $R
python3 scratch/pmc_summary.py gpurun_out "k_" ${TAG}+stage2+ > gpurun_out/${TAG}_pmc_stage2.txt
if [ "$2" = "extras" ]; then
  : > gpurun_out/${TAG}_pmc_extras.txt
  for W in stage1 stage1_qat q88 gen_pa gen_pa_float gen_pa_train spectral; do
    python3 scratch/pmc_summary.py gpurun_out "k_" ${TAG}+${W}+ >> gpurun_out/${TAG}_pmc_extras.txt
  done
fi
echo collected
