"""Which fused plan a shape takes: python scripts/dev/plans.py hinge:4096x300x64 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pytorchltr_amd import _C
torch.zeros(1, device="cuda:0")
names = {0: "none", 1: "regtile", 2: "cluster", 3: "general", 4: "parts"}
for case in sys.argv[1:]:
    kind, shp = case.split(":")
    B, L, F = (int(v) for v in shp.split("x"))
    print(case, names[_C.lib().ltr_linear_fused_plan(getattr(_C, kind.upper()), B, L, F)])
