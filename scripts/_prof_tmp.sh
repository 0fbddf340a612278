set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o c2 --output-format rocpd -- python $R/bench.py --steps 200 --warmup 20 > $R/gpurun_out/prof/c2_bench.json 2> $R/gpurun_out/prof/c2_err.log
ls -R /tmp/prof_c2 | head
DB=$(find /tmp/prof_c2 -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py $DB > $R/gpurun_out/prof/r01_c2_trace.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -o bench --output-format rocpd -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
python $R/scripts/rocpd_summary.py $(find /tmp/p_fetch -name "*.db" | head -1) > $R/gpurun_out/prof/r01_c2_fetch.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_write -o bench --output-format rocpd -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
python $R/scripts/rocpd_summary.py $(find /tmp/p_write -name "*.db" | head -1) > $R/gpurun_out/prof/r01_c2_write.txt
