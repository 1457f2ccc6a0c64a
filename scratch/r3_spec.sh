#!/bin/bash
# spectral kernels: tests, then kernel durations under rocprofv3 for the product build and a variant
R=$PWD
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_spectral_quant.py -x -q --timeout 200 --timeout-method thread -k spectral > gpurun_out/r3_t3.log 2>&1
rc=$?; tail -n 4 gpurun_out/r3_t3.log; [ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
for v in product "$@"; do
  if [ $v = product ]; then unset DPD_HIP_LIBRARY; else export DPD_HIP_LIBRARY=$R/scratch/lib_$v.so; fi
  rm -rf $R/gpurun_out/spec_$v
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/spec_$v -o s --output-format csv -- python3 $R/scratch/tk.py spec reps=30 > $R/gpurun_out/spec_$v.log 2>&1 || { echo "$v failed"; tail -n 5 $R/gpurun_out/spec_$v.log; exit 1; }
  echo "== $v"; grep -E "k_stats|k_bwd" $R/gpurun_out/spec_$v/s_kernel_stats.csv | sed -E 's/^"void \(anonymous namespace\):://; s/\(.*\)",/ /' | cut -c1-100
done
