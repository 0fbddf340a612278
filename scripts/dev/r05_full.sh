mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05/t_all.log
timeout 900 python bench.py > gpurun_out/r05/bench.json 2> gpurun_out/r05/bench.err
tail -3 gpurun_out/r05/t_all.log; tail -c 600 gpurun_out/r05/bench.err
