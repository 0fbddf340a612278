#!/bin/bash
# round 4, session 3: the NDCG kinds on the parts kernel -- tests, then parts vs general (LTR_DISABLE_PARTS=1) timing
mkdir -p gpurun_out/s3
python -m pytest tests/test_gpu_fullsize.py -x -q -k "parts_kernel_shapes" > gpurun_out/s3/ndcg_tests.log 2>&1
tail -n 5 gpurun_out/s3/ndcg_tests.log
for w in c5; do
  python scripts/time_fused_kinds.py pytorchltr_amd/csrc/libltr_hip.so --workload $w --kinds hinge,logistic,ndcg1,ndcg2 > gpurun_out/s3/kinds_time_parts_$w.log 2>&1
  tail -n 1 gpurun_out/s3/kinds_time_parts_$w.log
done
