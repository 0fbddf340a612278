#!/bin/bash
set -u
OUT=gpurun_out/r04
mkdir -p $OUT
L=$OUT/run4.log
echo "== sweep" > $L
for w in 2 3 4; do for rt in 24 32 48 64 96 128; do
  echo "== wpc $w rt $rt" >> $L
  LTR_PARTS_NOSORT=1 LTR_PARTS_DEBUG=1 LTR_PARTS_WPC=$w LTR_PARTS_RT=$rt timeout 120 python scripts/dev/parts_check.py --time --nocheck --shapes 32,1000,220,dcg_hinge 64,512,700,hinge 256,1000,220,dcg_hinge 6,1000,220,dcg_hinge 2>&1 | grep -v amdgpu.ids | sort | uniq >> $L
done; done
cat $L | cut -c1-200
