mkdir -p gpurun_out/r05
V=build/variants
timeout 400 python scripts/dev/lib_ab.py pytorchltr_amd/csrc/libltr_hip.so $V/libltr_wide.so -- hinge:64x512x700 hinge:32x512x700 dcg_hinge:64x1000x700 hinge:16x512x700 hinge:64x512x520 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/ab7.log 2>&1
cat gpurun_out/r05/ab7.log
