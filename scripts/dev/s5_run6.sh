mkdir -p gpurun_out/s5
cases="hinge:512x512x700 dcg_hinge:512x512x700 logistic:512x512x700 hinge:1024x512x700 hinge:256x1000x700 ndcg2:512x512x700 hinge:2048x128x700"
timeout 900 python scripts/dev/lib_ab.py build/variants/libltr_base.so pytorchltr_amd/csrc/libltr_hip.so -- $cases 2>&1 | grep -v amdgpu.ids > gpurun_out/s5/ab7.log
cat gpurun_out/s5/ab7.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/s5/t7.log
