#!/bin/bash
set -u
OUT=gpurun_out/r04
mkdir -p $OUT
L=$OUT/run2.log
CK="6,512,700,hinge 6,1000,220,dcg_hinge 40,300,64,logistic 33,600,136,arp1 20,700,220,arp2 64,512,700,hinge 32,1000,220,dcg_hinge 256,1000,220,dcg_hinge 512,512,700,hinge 100,1000,220,logistic 300,400,136,arp2 7,1024,64,hinge"
SH="256,1000,220,dcg_hinge 32,1000,220,dcg_hinge 64,512,700,hinge 512,512,700,hinge 128,600,136,hinge 256,1000,220,logistic"
echo "== check auto" > $L
LTR_PARTS_DEBUG=1 timeout 600 python scripts/dev/parts_check.py --time --shapes $CK >> $L 2>&1
for w in 2 3 4; do
  echo "== sym wpc$w" >> $L
  LTR_PARTS_WPC=$w LTR_PARTS_DEBUG=1 timeout 300 python scripts/dev/parts_check.py --time --nocheck --shapes $SH >> $L 2>&1
done
echo "== sym wpc4 nodirect" >> $L
LTR_PARTS_WPC=4 LTR_PARTS_NODIRECT=1 timeout 300 python scripts/dev/parts_check.py --time --nocheck --shapes $SH >> $L 2>&1
grep -v amdgpu.ids $L | cut -c1-400
