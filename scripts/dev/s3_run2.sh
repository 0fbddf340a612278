#!/bin/bash
mkdir -p gpurun_out/s3
S2="96x512x700 128x512x700 192x512x700 256x512x700 256x400x700 128x768x700 256x768x700 192x400x700 256x1000x700 160x1000x512"
LTR_PARTS_ALL=1 python scripts/dev/time_shapes.py --kinds hinge,logistic,arp1 $S2 > gpurun_out/s3/cold4_parts.log 2>&1
LTR_DISABLE_PARTS=1 python scripts/dev/time_shapes.py --kinds hinge,logistic,arp1 $S2 > gpurun_out/s3/cold4_general.log 2>&1
for f in cold4_parts cold4_general; do grep -v amdgpu.ids gpurun_out/s3/$f.log; echo; done
