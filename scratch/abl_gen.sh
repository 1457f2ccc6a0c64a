#!/bin/bash
# kernel times of the generator backward kernels for experiment builds scratch/lib_abl_*.so
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in BASE "$@"; do
  if [ $v = BASE ]; then unset DPD_HIP_LIBRARY; else export DPD_HIP_LIBRARY=$R/scratch/lib_abl_$v.so; fi
  for W in stage1 stage1_qat gen_pa_train; do
    timeout -k 10 120 python3 $R/bench.py --workload $W --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/abl_g_${v}_$W.json 2>$R/gpurun_out/abl_g_${v}_$W.err || { echo "$v $W failed"; tail -n 5 $R/gpurun_out/abl_g_${v}_$W.err; exit 1; }
    python3 -c "
import json,sys
d=json.loads([l for l in open('$R/gpurun_out/abl_g_${v}_$W.json') if l.startswith('{')][-1])
print('$v $W ms_per_step %.4f kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done
