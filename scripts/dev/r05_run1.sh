mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -k "cluster or sorted_runs" 2>&1 | tail -5 > gpurun_out/r05/t_cluster.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "c4 or timeout or shard" 2>&1 | tail -5 >> gpurun_out/r05/t_cluster.log
V=build/variants
timeout 400 python scripts/dev/lib_ab.py $V/libltr_base.so pytorchltr_amd/csrc/libltr_hip.so -- dcg_hinge:32x1000x220 hinge:64x512x700 dcg_hinge:256x1000x220 hinge:128x600x136 hinge:128x1000x220 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/ab1.log 2>&1
for shp in "32 1000 220 1"; do
echo "== $shp"; timeout 120 python scripts/trace_cluster.py $shp 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05/trace3.log 2>&1
cat gpurun_out/r05/t_cluster.log gpurun_out/r05/ab1.log gpurun_out/r05/trace3.log
