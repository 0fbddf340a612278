#!/bin/bash
mkdir -p gpurun_out/s3
python -m pytest tests/test_gpu_fused.py -x -q > gpurun_out/s3/gputest8.log 2>&1; tail -n 3 gpurun_out/s3/gputest8.log
S="1024x160x136 1024x200x136 1024x256x136 2048x200x136 512x256x136 1024x128x220 1024x160x220 1024x200x220 1024x216x220 256x256x136"
python scripts/dev/time_shapes.py --kinds hinge,logistic,arp1 $S > gpurun_out/s3/rt1024_new.log 2>&1
LTR_NO_REGTILE1024=1 python scripts/dev/time_shapes.py --kinds hinge,logistic,arp1 $S > gpurun_out/s3/rt1024_old.log 2>&1
for f in rt1024_new rt1024_old; do echo "## $f"; grep -v amdgpu.ids gpurun_out/s3/$f.log; done
