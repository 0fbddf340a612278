#!/bin/bash
mkdir -p gpurun_out/s3
python -m pytest tests -m gpu -x -q > gpurun_out/s3/gputest2.log 2>&1; tail -n 4 gpurun_out/s3/gputest2.log
S2="256x300x700 512x512x448 160x1000x512 256x512x700 512x512x700 384x1000x512 1024x512x512"
python scripts/dev/time_shapes.py --kinds hinge,logistic,arp1,ndcg2 $S2 > gpurun_out/s3/cold5_default.log 2>&1
grep -v amdgpu.ids gpurun_out/s3/cold5_default.log
