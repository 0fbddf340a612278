#!/bin/bash
set -u
OUT=gpurun_out/r04
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1
tail -15 $OUT/gputests.log
SH="512,512,700,hinge 1024,512,700,hinge 300,512,700,logistic 512,2000,64,hinge 256,4000,32,hinge 64,512,700,hinge 256,1000,220,dcg_hinge 32,1000,220,dcg_hinge"
for w in 2 3; do
echo "== parts wpc$w (auto plans)" >> $OUT/run6.log
LTR_PARTS_WPC=$w timeout 300 python scripts/dev/parts_check.py --time --nocheck --shapes $SH 2>&1 | grep kernel_us >> $OUT/run6.log
done
cat $OUT/run6.log
