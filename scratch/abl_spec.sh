#!/bin/bash
# kernel statistics of the SpectralLoss workload for experiment builds scratch/lib_abl_*.so
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in BASE "$@"; do
  if [ $v = BASE ]; then unset DPD_HIP_LIBRARY; else export DPD_HIP_LIBRARY=$R/scratch/lib_abl_$v.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abl_sp_$v -o s --output-format csv -- python3 $R/bench.py --workload spectral --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $R/gpurun_out/abl_sp_$v.log 2>&1 || { echo "$v failed"; tail -n 5 $R/gpurun_out/abl_sp_$v.log; exit 1; }
  echo "== $v"; grep -E "k_stats|k_bwd|k_rowdb|k_final|k_terms" $R/gpurun_out/abl_sp_$v/s_kernel_stats.csv | sed -E 's/^"(void )?\(anonymous namespace\):://; s/\(.*\)",/ /' | cut -c1-80
done
