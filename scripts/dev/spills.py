"""Register / spill census of the parts kernel without a full build: the linear translation unit compiled
device-only with -DLTR_DEV_SUBSET (hinge + logistic, symmetric split pass only) and
-Rpass-analysis=kernel-resource-usage.   python scripts/dev/spills.py [extra hipcc flags]"""
import re, subprocess, sys
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-pthread", "-Wno-unused-function",
       "-Wno-bitwise-instead-of-logical", "-I", "include", "-I", "pytorchltr_amd/csrc", "--cuda-device-only", "-c",
       "pytorchltr_amd/csrc/ltr_linear.hip", "-o", "/tmp/x%d.o" % __import__("os").getpid(), "-Rpass-analysis=kernel-resource-usage"] + ([] if "--full" in sys.argv else ["-DLTR_DEV_SUBSET"]) + [a for a in sys.argv[1:] if a != "--full"]
out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE).stderr.decode()
name = None
rows = []
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m: name = m.group(1); cur = {"name": name}; rows.append(cur)
    for key in ("VGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize \\[bytes/lane\\]"):
        m = re.search(r"remark:\s+%s: (\d+)" % key, ln)
        if m and rows: rows[-1][key] = int(m.group(1))
    if "error" in ln: print(ln)
for r in rows:
    if "linear_parts_kernel" in r["name"]:
        t = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", r["name"]).groups()
        print("kind %s ni %2s cv %s wpc %s mode %s: vgpr %3s spill %3s" % (*t, r.get("VGPRs"), r.get("VGPRs Spill")))
