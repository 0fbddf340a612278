#!/bin/bash
# Build tuning variants of the library into build/variants/: the fused-linear translation unit (ltr_linear.hip:
# register tile, cluster and parts kernels) recompiled with the given flags and linked with the other units'
# objects of the last full build (python -m pytorchltr_amd.build).
# usage: scripts/build_variants.sh name1:"-DX=1 -DY=2" name2:"..."
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
C="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -pthread -Wno-unused-function -I include -I pytorchltr_amd/csrc"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( $C $flags -c pytorchltr_amd/csrc/ltr_linear.hip -o build/variants/ltr_linear_$name.o &&
    $C -shared -o build/variants/libltr_$name.so build/obj/ltr_kernels.o build/variants/ltr_linear_$name.o build/obj/ltr_mlp.o ) &
done
wait
ls -la build/variants/*.so
