# the round's evidence, collected once on the final build: kernel trace of the default command, PMC passes c2..c5, the default bench
# line, the lazy launch and the drop-in loop under the profiler, phase traces, the per-CU byte histogram, the B-sweep, the
# data-parallel stress on one GPU
mkdir -p gpurun_out/r06f
timeout 2700 bash scripts/collect_profiles.sh r06f c2 c3 c4 c5 > gpurun_out/r06f/collect.log 2>&1
timeout 900 python bench.py > gpurun_out/r06f/bench.json 2> gpurun_out/r06f/bench.err
tail -c 300 gpurun_out/r06f/bench.err
scripts/pmc_pass.sh gpurun_out/r06f/prof/lazy none -- python $PWD/scripts/lazy_trace.py lazy 3000
scripts/pmc_pass.sh gpurun_out/r06f/prof/eager none -- python $PWD/scripts/lazy_trace.py eager 3000
scripts/pmc_pass.sh gpurun_out/r06f/prof/dropin_lazy none -- python $PWD/scripts/dropin_loop.py lazy 1000
scripts/pmc_pass.sh gpurun_out/r06f/prof/dropin_torch none -- python $PWD/scripts/dropin_loop.py torch 1000
( python scripts/trace_regtile.py build/variants/libltr_trace.so --workload c3; python scripts/trace_regtile.py build/variants/libltr_trace.so --workload c2 ) 2>&1 | grep -v amdgpu > gpurun_out/r06f/regtile_phases.txt
python scripts/cu_bytes.py build/variants/libltr_trace.so 2>&1 | grep launch > gpurun_out/r06f/cu_bytes.txt
( LTR_TRACE_LIB=$PWD/build/variants/libltr_trace.so python scripts/trace_cluster.py 256 1000 220 1; LTR_TRACE_LIB=$PWD/build/variants/libltr_trace.so python scripts/trace_cluster.py 32 1000 220 1 ) 2>&1 | grep -v amdgpu > gpurun_out/r06f/cluster_phases.txt
timeout 1200 python scripts/sweep_b.py 2>&1 | grep -v amdgpu > gpurun_out/r06f/sweep_b.jsonl
( LTR_MAILBOX_TIMEOUT_MS=5000 timeout 300 python scripts/dp_stress.py 2 3000; DP_SHARDS=16,64,96 LTR_MAILBOX_TIMEOUT_MS=5000 timeout 300 python scripts/dp_stress.py 8 2000 ) 2>&1 | grep '^{' > gpurun_out/r06f/dp_stress.jsonl
ls gpurun_out/r06f gpurun_out/r06f/prof | head -60
