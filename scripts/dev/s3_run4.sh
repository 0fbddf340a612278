#!/bin/bash
mkdir -p gpurun_out/s3
S="128x512x136 256x512x136 128x1000x136 256x1000x136 192x1000x136 128x512x220 128x1000x220 256x1000x220 192x1000x220 192x700x220 160x1000x512 200x768x512 128x700x448 256x1000x700 512x512x700 96x1000x136 256x700x136"
LTR_PARTS_FIRST=1 LTR_PARTS_ALL=1 python scripts/dev/time_shapes.py --kinds ndcg1,ndcg2 $S > gpurun_out/s3/audit_ndcg_parts2.log 2>&1
LTR_DISABLE_PARTS=1 python scripts/dev/time_shapes.py --kinds ndcg1,ndcg2 192x1000x136 192x1000x220 192x700x220 96x1000x136 256x700x136 > gpurun_out/s3/audit_ndcg_general2.log 2>&1
for f in audit_ndcg_parts2 audit_ndcg_general2; do echo "## $f"; grep -v amdgpu.ids gpurun_out/s3/$f.log; echo; done
