#!/bin/bash
# VALU / SALU / LDS instruction counts of the register-tile kernel cut off after each phase
# (build/variants/libltr_stop{1,2,3}.so = -DLTR_V2_STOP=1..3, plus the product library)
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
cd "$root"
for kind in "$@"; do
  for v in stop1 stop2 stop3 full; do
    lib=build/variants/libltr_$v.so; [ $v = full ] && lib=pytorchltr_amd/csrc/libltr_hip.so
    scripts/pmc_pass.sh gpurun_out/r03/phase/${kind}_$v "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" -- python $root/scripts/run_kernel_only.py $root/$lib --workload c2 --kind $kind
    echo "== $kind $v"; grep -i "regtile" gpurun_out/r03/phase/${kind}_$v/summary.txt | head -3
  done
done
