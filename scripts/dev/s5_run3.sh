mkdir -p gpurun_out/s5
cases="hinge:512x512x700 dcg_hinge:512x512x700 hinge:1024x512x700 hinge:320x768x700 hinge:1024x300x700 hinge:2048x128x700 hinge:512x512x640 hinge:256x1000x700 hinge:768x400x576"
LTR_PARTS_RUNS=1 timeout 900 python scripts/dev/lib_ab.py build/variants/libltr_base.so pytorchltr_amd/csrc/libltr_hip.so -- $cases 2>&1 | grep -v amdgpu.ids > gpurun_out/s5/ab3.log
cat gpurun_out/s5/ab3.log
LTR_PARTS_RUNS=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c5 or parts or exchange" 2>&1 | tail -15 | tee gpurun_out/s5/t3.log
