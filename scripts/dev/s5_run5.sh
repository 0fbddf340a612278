mkdir -p gpurun_out/s5
cases="hinge:1024x128x136 ndcg2:1024x128x136 dcg_hinge:256x1000x220 dcg_hinge:32x1000x220 hinge:512x512x700 hinge:64x512x700 logistic:512x512x700 hinge:1024x256x220 hinge:512x1000x220 ndcg2:512x600x136"
timeout 900 python scripts/dev/lib_ab.py build/variants/libltr_base.so pytorchltr_amd/csrc/libltr_hip.so -- $cases 2>&1 | grep -v amdgpu.ids > gpurun_out/s5/ab6.log
cat gpurun_out/s5/ab6.log
