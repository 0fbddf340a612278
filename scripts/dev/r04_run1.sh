#!/bin/bash
# round 4, first GPU pass: the symmetric split pass in the parts kernel -- parity on every row, then timing
# against the cluster kernel and the one-sided parts kernel on the long-list shapes
set -u
OUT=gpurun_out/r04
mkdir -p $OUT
CK="6,512,700,hinge 6,1000,220,dcg_hinge 40,300,64,logistic 33,600,136,arp1 20,700,220,arp2 64,512,700,hinge 32,1000,220,dcg_hinge 256,1000,220,dcg_hinge 512,512,700,hinge 100,1000,220,logistic"
SH="256,1000,220,dcg_hinge 32,1000,220,dcg_hinge 64,512,700,hinge 512,512,700,hinge 128,600,136,hinge 256,1000,220,logistic"
echo "== check auto" > $OUT/run1.log
timeout 600 python scripts/dev/parts_check.py --time --shapes $CK >> $OUT/run1.log 2>&1
for w in 2 3 4; do
  echo "== sym wpc$w" >> $OUT/run1.log
  LTR_PARTS_WPC=$w LTR_PARTS_DEBUG=1 timeout 300 python scripts/dev/parts_check.py --time --nocheck --shapes $SH >> $OUT/run1.log 2>&1
done
echo "== cluster (old)" >> $OUT/run1.log
LTR_USE_CLUSTER=1 timeout 300 python scripts/dev/parts_check.py --time --nocheck --shapes $SH >> $OUT/run1.log 2>&1
echo "== one-sided parts (old), all shapes" >> $OUT/run1.log
LTR_PARTS_NOSYM=1 LTR_PARTS_ALL=1 timeout 300 python scripts/dev/parts_check.py --time --nocheck --shapes $SH >> $OUT/run1.log 2>&1
tail -80 $OUT/run1.log
