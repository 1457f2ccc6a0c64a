#!/bin/bash
# kernel times of GP pass A with and without its global loads / stores (experiment builds in scratch/lib_abl_*.so)
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in BASE NOSTORE NOLOAD BOTH; do
  if [ $v = BASE ]; then unset DPD_HIP_LIBRARY; else export DPD_HIP_LIBRARY=$R/scratch/lib_abl_$v.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abl_gpa_$v -o s --output-format csv -- python3 $R/bench.py --workload stage2 --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/abl_gpa_$v.log 2>&1 || { echo "$v failed"; tail -n 5 $R/gpurun_out/abl_gpa_$v.log; exit 1; }
  echo "== $v"; grep -E "k_gp_a_mfma|k_gp_c_mfma|k_critic_bwd_mfma<false" $R/gpurun_out/abl_gpa_$v/s_kernel_stats.csv | awk -F'",' '{n=split($1,a,"("); print a[1], $2}' | cut -c1-120
done
