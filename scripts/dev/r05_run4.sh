mkdir -p gpurun_out/r05
for shp in "256 1000 220 1" "32 1000 220 1"; do
echo "== $shp"; timeout 120 python scripts/trace_cluster.py $shp 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05/trace4.log 2>&1
cat gpurun_out/r05/trace4.log
