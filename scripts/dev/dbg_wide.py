import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from tests.conftest import synth
from pytorchltr_amd import _C
dev = torch.device("cuda:0")
B, L, F = 5, 8, 2048
s, y, n, X, W, b = synth(B, L, 17, F=F)
X, W, b, y, n = X.to(dev), W.to(dev), b.to(dev), y.to(dev), n.to(dev)
lib = _C.lib()
ws = torch.full((lib.ltr_linear_workspace_bytes(B, L, F) // 4,), 7.0, device=dev)
loss = torch.empty(B, device=dev)
rc = lib.ltr_linear_partials_f32(_C.HINGE, 1.0, X.data_ptr(), W.data_ptr(), b.data_ptr(), y.data_ptr(), 0, n.data_ptr(),
                                 B, L, F, loss.data_ptr(), None, ws.data_ptr(), _C.stream_of(X))
torch.cuda.synchronize()
PF = (F + 4) & ~3
part = ws[:B * PF].reshape(B, PF).cpu().numpy()
print("rc", rc, "n", n.cpu().numpy(), "loss", loss.cpu().numpy())
print("bias col", part[:, F], "pad", part[:, F + 1:F + 4])
print("col0", part[:, 0], "col F-1", part[:, F - 1])
