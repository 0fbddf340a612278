mkdir -p gpurun_out/r05
cases=""
for F in 136 220; do for L in 800 900 1000; do for B in 160 192 224 256 320 384; do cases="$cases hinge:${B}x${L}x${F}"; done; done; done
for lb in 0 100000; do
echo "LIGHT_MAXB=$lb"
LTR_CLUSTER_LIGHT_MAXB=$lb timeout 800 python scripts/dev/lib_ab.py pytorchltr_amd/csrc/libltr_hip.so -- $cases 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05/ab8.log 2>&1
python - <<'PY'
import re
d={}
cur=None
for ln in open('gpurun_out/r05/ab8.log'):
    if ln.startswith("LIGHT"): cur=ln.strip(); continue
    m=re.match(r"(\S+) \| hip ([\d.]+)/([\d.]+)/", ln)
    if m: d.setdefault(m.group(1),{})[cur]=float(m.group(3))
for k,v in d.items():
    a=v.get("LIGHT_MAXB=0"); b=v.get("LIGHT_MAXB=100000")
    print(k, a, b, "%+.1f%%" % ((b/a-1)*100) if a and b else "")
PY
