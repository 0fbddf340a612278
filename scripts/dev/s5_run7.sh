mkdir -p gpurun_out/s5
cases="dcg_hinge:256x1000x220 dcg_hinge:32x1000x220 hinge:64x512x700 hinge:256x1000x136 hinge:128x600x136 hinge:64x1000x700 hinge:128x1000x220 hinge:192x800x220 hinge:100x1000x64"
timeout 900 python scripts/dev/lib_ab.py build/variants/libltr_base.so pytorchltr_amd/csrc/libltr_hip.so -- $cases 2>&1 | grep -v amdgpu.ids > gpurun_out/s5/ab8.log
cat gpurun_out/s5/ab8.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/s5/t8.log
