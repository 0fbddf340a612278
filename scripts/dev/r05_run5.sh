mkdir -p gpurun_out/r05
V=build/variants
for lb in 0 100000; do
echo "LIGHT_MAXB=$lb"
LTR_CLUSTER_LIGHT_MAXB=$lb timeout 400 python scripts/dev/lib_ab.py pytorchltr_amd/csrc/libltr_hip.so -- dcg_hinge:256x1000x220 hinge:128x1000x220 dcg_hinge:128x1000x220 hinge:256x700x220 hinge:192x600x136 hinge:256x1000x136 logistic:256x1000x220 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05/ab5.log 2>&1
cat gpurun_out/r05/ab5.log
