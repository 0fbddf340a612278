#!/bin/bash
mkdir -p gpurun_out/s3
S="200x768x512 256x768x512 224x1000x512 256x1000x640 320x768x700 200x600x448 256x600x512 192x512x512"
python scripts/dev/time_shapes.py --kinds ndcg1,ndcg2 $S > gpurun_out/s3/nd_a.log 2>&1
LTR_PARTS_FIRST=1 LTR_PARTS_ALL=1 python scripts/dev/time_shapes.py --kinds ndcg1,ndcg2 $S > gpurun_out/s3/nd_b.log 2>&1
for f in nd_a nd_b; do echo "## $f"; grep -v amdgpu.ids gpurun_out/s3/$f.log; done
