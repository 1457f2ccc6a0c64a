#!/bin/bash
script=$1; shift
for v in "$@"; do
  DPD_HIP_LIBRARY=$PWD/scratch/lib_$v.so timeout -k 5 120 python scratch/$script 30 || exit 1
done
