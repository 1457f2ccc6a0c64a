#!/bin/bash
# kernel times of the critic kernels with extra independent instructions per tile (experiment builds scratch/lib_abl_*.so)
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in BASE "$@"; do
  if [ $v = BASE ]; then unset DPD_HIP_LIBRARY; else export DPD_HIP_LIBRARY=$R/scratch/lib_abl_$v.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abl_x_$v -o s --output-format csv -- python3 $R/bench.py --workload stage2 --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/abl_x_$v.log 2>&1 || { echo "$v failed"; tail -n 5 $R/gpurun_out/abl_x_$v.log; exit 1; }
  echo "== $v"; grep -E "k_gp_|k_critic_bwd_mfma" $R/gpurun_out/abl_x_$v/s_kernel_stats.csv | sed -E 's/^"void \(anonymous namespace\):://; s/\(.*\)",/ /' | cut -c1-90
done
