#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
for spec in "256 1000 220 1" "32 1000 220 1" "64 512 700 0"; do
  for w in 2 3; do
    echo "== trace $spec wpc $w" >> $OUT/trace1.log
    LTR_PARTS_WPC=$w timeout 120 python scripts/dev/trace_sparts.py $spec >> $OUT/trace1.log 2>&1
  done
done
cat $OUT/trace1.log
