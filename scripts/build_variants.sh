#!/bin/bash
# Build tuning variants of the extension (hinge-only fast builds) into build/variants/.
# usage: scripts/build_variants.sh name1:"-DX=1 -DY=2" name2:"..."
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc \
     -Wno-unused-function -I include -I pytorchltr_amd/csrc $flags \
     -o build/variants/libltr_$name.so pytorchltr_amd/csrc/ltr_kernels.hip &
done
wait
ls -la build/variants/
