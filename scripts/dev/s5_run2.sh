mkdir -p gpurun_out/s5
for r in 0 1; do echo "== LTR_PARTS_RUNS=$r"; LTR_PARTS_RUNS=$r timeout 300 python scripts/dev/trace_parts.py 512 512 700 0 2>&1 | grep -v amdgpu.ids; done > gpurun_out/s5/tr2.log
cat gpurun_out/s5/tr2.log
