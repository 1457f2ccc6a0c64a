#!/bin/bash
# kernel times of the generator kernels for experiment builds scratch/lib_abl_*.so (rocprofv3 kernel statistics)
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in BASE "$@"; do
  if [ $v = BASE ]; then unset DPD_HIP_LIBRARY; else export DPD_HIP_LIBRARY=$R/scratch/lib_abl_$v.so; fi
  for W in stage1 gen_pa gen_pa_float gen_pa_train; do
    timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abl_g2_${v}_$W -o s --output-format csv -- python3 $R/bench.py --workload $W --no-extras --no-cpu-baseline --steps 5 --warmup 2 > $R/gpurun_out/abl_g2_${v}_$W.log 2>&1 || { echo "$v $W failed"; tail -n 5 $R/gpurun_out/abl_g2_${v}_$W.log; exit 1; }
    echo "== $v $W"; grep -E "k_gen_|k_iq_absmax|k_pa_|k_quantize" $R/gpurun_out/abl_g2_${v}_$W/s_kernel_stats.csv | sed -E 's/^"(void )?\(anonymous namespace\):://; s/\(.*\)",/ /' | cut -c1-80
  done
done
