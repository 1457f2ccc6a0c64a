#!/bin/bash
# usage: scratch/build_variants.sh NAME:FLAGS ...   builds scratch/lib_NAME.so (gen_mfma.hip recompiled with FLAGS)
set -e
cd "$(dirname "$0")/.."
PK=6g-pa-gan-dpd_amd
OBJS=$(ls $PK/build/*.o | grep -v gen_mfma)
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-result -I include $flags -c $PK/csrc/gen_mfma.hip -o scratch/gm_$name.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS scratch/gm_$name.o -o scratch/lib_$name.so && echo built $name ) &
done
wait
