import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/scripts')
import tune
from _benchutil import time_launches
from pytorchltr_amd import _C
dev=torch.device('cuda:0')
for (B,L,F) in ((1024,200,136),(1024,600,32),(512,300,64),(1024,128,136)):
    g=torch.Generator().manual_seed(0)
    rel=torch.randint(0,5,(B,L),generator=g).to(dev); n=torch.randint(1,L+1,(B,),generator=g).to(dev)
    X=torch.randn(B,L,F,device=dev); W=torch.randn(F,device=dev)*0.1; bias=torch.zeros(1,device=dev)
    loss=torch.empty(B,device=dev)
    out=[]
    for path in sys.argv[1:]:
        lib=tune.load(path)
        part=torch.empty(lib.ltr_linear_workspace_bytes(B,L,F)//4,device=dev)
        def fused():
            rc=lib.ltr_linear_partials_f32(0,1.0,X.data_ptr(),W.data_ptr(),bias.data_ptr(),rel.data_ptr(),0,n.data_ptr(),B,L,F,loss.data_ptr(),None,part.data_ptr(),torch.cuda.current_stream().cuda_stream); assert rc==0
        for _ in range(5): fused()
        t,_=time_launches(fused,per_graph=10,replays=5)
        out.append("%s %.1f us" % (path.split('/')[-1], t))
    print((B,L,F), " | ".join(out), flush=True)
