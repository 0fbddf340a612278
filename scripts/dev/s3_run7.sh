#!/bin/bash
mkdir -p gpurun_out/s3
S="1024x128x700 2048x128x700 512x128x700 1024x200x700 1024x256x700 1024x128x512 1024x200x448 512x256x640 256x256x700"
LTR_PARTS_ALL=1 LTR_PARTS_FIRST=1 python scripts/dev/time_shapes.py --kinds hinge,logistic,ndcg2 $S > gpurun_out/s3/short_parts.log 2>&1
for f in short_parts; do echo "## $f"; grep -v amdgpu.ids gpurun_out/s3/$f.log; done
