#!/bin/bash
mkdir -p gpurun_out/s3
python scripts/dev/dbg_ndcg_rt.py 2>&1 | grep -v amdgpu.ids
S="1024x160x136 1024x200x136 1024x256x136 512x256x136 2048x200x136 1024x160x220"
python scripts/dev/time_shapes.py --kinds ndcg1,ndcg2 $S > gpurun_out/s3/rtn_new.log 2>&1
LTR_NO_REGTILE1024_NDCG=1 python scripts/dev/time_shapes.py --kinds ndcg1,ndcg2 $S > gpurun_out/s3/rtn_old.log 2>&1
for f in rtn_new rtn_old; do echo "## $f"; grep -v amdgpu.ids gpurun_out/s3/$f.log; done
