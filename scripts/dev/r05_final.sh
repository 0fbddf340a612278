# the round's evidence, collected once on the final build: profiles (kernel trace of the default command, PMC passes c2..c5), the default bench line
mkdir -p gpurun_out/r05f
timeout 2700 bash scripts/collect_profiles.sh r05f c2 c3 c4 c5 > gpurun_out/r05f/collect.log 2>&1
timeout 900 python bench.py > gpurun_out/r05f/bench.json 2> gpurun_out/r05f/bench.err
tail -c 300 gpurun_out/r05f/bench.err
ls gpurun_out/r05f/prof | head -30
