#!/bin/bash
set -u
OUT=gpurun_out/r04
mkdir -p $OUT
L=$OUT/run3.log
CK="6,512,700,hinge 6,1000,220,dcg_hinge 33,600,136,hinge 20,700,220,dcg_hinge 64,512,700,hinge 32,1000,220,dcg_hinge 256,1000,220,dcg_hinge 512,512,700,hinge 100,1000,220,logistic 7,1024,64,hinge 50,300,64,hinge 40,520,136,dcg_hinge 9,97,700,hinge"
SH="256,1000,220,dcg_hinge 32,1000,220,dcg_hinge 64,512,700,hinge 512,512,700,hinge 128,600,136,hinge 256,1000,220,logistic"
echo "== check auto" > $L
LTR_PARTS_ALL=1 timeout 600 python scripts/dev/parts_check.py --time --shapes $CK >> $L 2>&1
for w in 2 3 4; do
  echo "== sym wpc$w" >> $L
  LTR_PARTS_WPC=$w timeout 300 python scripts/dev/parts_check.py --time --nocheck --shapes $SH >> $L 2>&1
done
echo "== nosort wpc3" >> $L
LTR_PARTS_WPC=3 LTR_PARTS_NOSORT=1 timeout 300 python scripts/dev/parts_check.py --time --nocheck --shapes $SH >> $L 2>&1
grep -v "amdgpu.ids\|^parts:" $L | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): print(l); continue
    try: d=json.loads(l)
    except Exception: print('??', l[:300]); continue
    if 'rep0' in d:
        ok=all(d['rep%d'%i]['loss_ok'] and d['rep%d'%i]['dW_err']<d['rep%d'%i]['tol'] and d['rep%d'%i]['bad_rows']==0 for i in range(3))
        bi=all(d['rep%d'%i].get('bit_identical',True) for i in range(3))
        print(d['shape'], 'plan',d['plan'],'OK' if ok else 'FAIL '+json.dumps(d['rep0']), 'bitid' if bi else 'NOT-BITID', 'status',d['status'], d.get('kernel_us'))
    else: print(d)
"
