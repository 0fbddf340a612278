#!/bin/bash
set -u
OUT=gpurun_out/r04
mkdir -p $OUT
L=$OUT/run5.log
SH="256,1000,220,dcg_hinge 32,1000,220,dcg_hinge 64,512,700,hinge 512,512,700,hinge 128,600,136,hinge 256,1000,220,logistic 6,1000,220,dcg_hinge 128,1000,220,hinge"
echo "== auto" > $L
LTR_PARTS_DEBUG=1 timeout 300 python scripts/dev/parts_check.py --time --nocheck --shapes $SH 2>&1 | grep -v amdgpu | uniq >> $L
for w in 2 3 4; do
  echo "== wpc$w" >> $L
  LTR_PARTS_WPC=$w timeout 300 python scripts/dev/parts_check.py --time --nocheck --shapes $SH 2>&1 | grep kernel_us >> $L
done
echo "== wpc3 nolight" >> $L
LTR_PARTS_WPC=3 LTR_PARTS_NOLIGHT=1 timeout 300 python scripts/dev/parts_check.py --time --nocheck --shapes $SH 2>&1 | grep kernel_us >> $L
echo "== cluster" >> $L
LTR_USE_CLUSTER=1 timeout 300 python scripts/dev/parts_check.py --time --nocheck --shapes $SH 2>&1 | grep kernel_us >> $L
echo "== check" >> $L
timeout 600 python scripts/dev/parts_check.py --shapes 6,1000,220,dcg_hinge 33,600,136,arp1 64,512,700,hinge 32,1000,220,dcg_hinge 256,1000,220,logistic 512,512,700,hinge 2>&1 | grep -v amdgpu >> $L
TRACE_OUT=trace3.log TRACE_SPECS="32,1000,220,1,4,64 256,1000,220,1,3,128" bash scripts/dev/r04_trace.sh > /dev/null 2>&1
cat $L | cut -c1-300
