#!/bin/bash
# usage: scratch/variant.sh FILE NAME "FLAGS"   ->  scratch/lib_NAME.so: csrc/FILE.hip recompiled with FLAGS, every other
# object taken from the product build (6g-pa-gan-dpd_amd/build/*.o).  Select a build with DPD_HIP_LIBRARY.
set -e
cd "$(dirname "$0")/.."
PK=6g-pa-gan-dpd_amd
file=$1; name=$2; flags=$3
OBJS=$(ls $PK/build/*.o | grep -v "/$file.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-result -I include $flags -c $PK/csrc/$file.hip -o scratch/v_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS scratch/v_$name.o -o scratch/lib_$name.so
echo built scratch/lib_$name.so
