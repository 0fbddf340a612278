#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
T=$OUT/${TRACE_OUT:-trace2.log}
rm -f $T
SPECS=${TRACE_SPECS:-"6,1000,220,1,3,64 32,1000,220,1,3,64 32,1000,220,1,4,64 256,1000,220,1,3,128 64,512,700,0,2,64"}
for spec in $SPECS; do
  IFS=, read B L F K W RT <<< "$spec"
  echo "== trace $B $L $F kind $K wpc $W rt $RT" >> $T
  LTR_PARTS_NOSORT=1 LTR_PARTS_WPC=$W LTR_PARTS_RT=$RT LTR_PARTS_DEBUG=1 timeout 120 python scripts/dev/trace_sparts.py $B $L $F $K 2>&1 | grep -v amdgpu.ids | uniq >> $T
done
cut -c1-900 $T
