#!/bin/bash
mkdir -p gpurun_out/s3
python -m pytest tests -m gpu -x -q > gpurun_out/s3/gputest5.log 2>&1; tail -n 3 gpurun_out/s3/gputest5.log
S="200x768x512 200x768x448 128x600x512 256x1000x512 160x768x640"
python scripts/dev/time_shapes.py --kinds hinge,logistic,ndcg1,ndcg2 $S > gpurun_out/s3/cl_a2.log 2>&1
grep -v amdgpu.ids gpurun_out/s3/cl_a2.log
