#!/bin/bash
set -u
OUT=gpurun_out/r04
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_bench.py -x -q > $OUT/steptests.log 2>&1
tail -25 $OUT/steptests.log
timeout 600 python bench.py > $OUT/bench_sgd.json 2> $OUT/bench_sgd.err
tail -3 $OUT/bench_sgd.err
python -c "
import json
d=json.load(open('$OUT/bench_sgd.json'))
print({k:d[k] for k in ('value','ms_per_step','n_gpus')}, d['roofline']['frac'], d['roofline']['kernel_us_avg'], d['extra'].get('step_without_weight_update'))
print(d['config']['step'])
for k,v in d['extra'].get('configs',{}).items(): print(k, v.get('plan'), v.get('kernel_us'), v.get('step_us'), v.get('frac_moved'))
"
