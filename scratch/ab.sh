#!/bin/bash
# kernel times of experiment builds next to the product build: ab.sh "<tk.py arguments>" NAME...   (scratch/lib_NAME.so)
args=$1; shift
for v in BASE "$@"; do
  if [ $v = BASE ]; then unset DPD_HIP_LIBRARY; else export DPD_HIP_LIBRARY=$PWD/scratch/lib_$v.so; fi
  timeout -k 5 100 python scratch/tk.py $args || exit 1
done
