#!/bin/bash
for v in nopk nomix; do
  echo "== $v"
  DPD_HIP_LIBRARY=$PWD/scratch/lib_$v.so timeout -k 10 120 python -u -m pytest tests/test_gpu_generator.py -x -q -k "float_generator_golden" --timeout 60 --timeout-method thread 2>&1 | tail -3
done
