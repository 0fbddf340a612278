mkdir -p gpurun_out/r05f/prof
for m in lazy eager; do scripts/pmc_pass.sh gpurun_out/r05f/prof/trace_$m none -- python $GRAFT_REPO_ROOT/scripts/lazy_trace.py $m 3000; head -8 gpurun_out/r05f/prof/trace_$m/summary.txt | cut -c1-170; done
