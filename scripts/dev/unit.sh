#!/bin/bash
# Recompile the named translation units (ltr_kernels / ltr_linear / ltr_mlp / ltr_steps) into build/obj and relink
# libltr_hip.so -- the development shortcut for "python -m pytorchltr_amd.build" when only one unit changed.
set -e
cd "$(dirname "$0")/../.."
C="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -pthread -Wall -Wno-unused-function -I include -I pytorchltr_amd/csrc"
for u in "$@"; do
  x=""; [ "$u" = ltr_steps ] && x="-mllvm -disable-machine-licm"
  $C $x $EXTRA -c pytorchltr_amd/csrc/$u.hip -o build/obj/$u.o &
done
wait
$C -shared -o pytorchltr_amd/csrc/libltr_hip.so build/obj/ltr_kernels.o build/obj/ltr_linear.o build/obj/ltr_mlp.o build/obj/ltr_steps.o
ls -la pytorchltr_amd/csrc/libltr_hip.so
