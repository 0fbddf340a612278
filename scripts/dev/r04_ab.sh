#!/bin/bash
# A/B of library variants through the default bench line:  scripts/dev/r04_ab.sh name=path ...
OUT=gpurun_out/r04
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  for rep in 1 2; do
    if [ "$lib" = "default" ]; then timeout 600 python bench.py --no-cpu-baseline > $OUT/ab_${name}_$rep.json 2>/dev/null;
    else LTR_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline > $OUT/ab_${name}_$rep.json 2>/dev/null; fi
    python - <<PY
import json
d=json.load(open('$OUT/ab_${name}_$rep.json'))
c=d['extra'].get('configs',{})
lk=d['extra'].get('loss_kernel',{})
print('$name', $rep, 'step %.2f kernel %.2f frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['kernel_us_avg'], d['roofline']['frac']),
      ' '.join('%s %.1f' % (k, v['kernel_us']) for k,v in c.items()), 'loss_kernel', {k: round(v,2) for k,v in lk.items() if isinstance(v,float) and 'us' in k})
PY
  done
done
