#!/bin/bash
# kernel times of the generator workloads for variant libraries scratch/lib_NAME.so next to the product build
timeout -k 10 120 python -u scratch/tk.py gen reps=20 2>/dev/null | grep -v amdgpu.ids || exit 1
for v in "$@"; do
  DPD_HIP_LIBRARY=$PWD/scratch/lib_$v.so timeout -k 10 120 python -u scratch/tk.py gen reps=20 2>/dev/null | grep "mse_step\|gen_fwd\|pa_fwd" || exit 1
done
