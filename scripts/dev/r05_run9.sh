mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py tests/test_gpu_step.py -x -q 2>&1 | tail -6 > gpurun_out/r05/t9.log
timeout 400 python scripts/dev/fuzz_dispatch.py 5 240 2>&1 | grep -v amdgpu | tail -8 >> gpurun_out/r05/t9.log
V=build/variants
timeout 400 python scripts/dev/lib_ab.py $V/libltr_base.so pytorchltr_amd/csrc/libltr_hip.so -- dcg_hinge:32x1000x220 hinge:64x512x700 dcg_hinge:256x1000x220 hinge:256x1000x136 hinge:128x1000x220 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05/t9.log 2>&1
cat gpurun_out/r05/t9.log
