#!/bin/bash
# ab2.sh "<tk.py arguments>" "OPTS" NAME...: as ab.sh with library options (DPD_OPTS) for every build
args=$1; export DPD_OPTS=$2; shift 2
for v in "$@"; do
  if [ $v = BASE ]; then unset DPD_HIP_LIBRARY; else export DPD_HIP_LIBRARY=$PWD/scratch/lib_$v.so; fi
  timeout -k 5 100 python scratch/tk.py $args || exit 1
done
