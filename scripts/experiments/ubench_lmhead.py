import os, sys
sys.path.insert(0, "/root/repo")
import torch
from spatialrgpt_amd import ops
dev="cuda"
N,K=128258,4096
Ws=[torch.randn((N,K),device=dev,dtype=torch.bfloat16)*0.02 for _ in range(3)]
x=torch.randn((1,K),device=dev,dtype=torch.bfloat16); g=torch.ones(K,device=dev,dtype=torch.bfloat16)
out=torch.empty((1,N),device=dev,dtype=torch.float32)
for W in Ws: ops.gemv(x,W,norm_w=g,eps=1e-5,out=out,out_f32=True)
torch.cuda.synchronize()
evs=[]
for _ in range(5):
    for W in Ws:
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemv(x,W,norm_w=g,eps=1e-5,out=out,out_f32=True); e1.record(); evs.append((e0,e1))
torch.cuda.synchronize()
ms=sorted(a.elapsed_time(b) for a,b in evs)
print(os.environ.get("SRGPT_GEMV_BLOCKS_PER_CU"), f"median {ms[len(ms)//2]*1e3:.1f} us  {N*K*2/ms[len(ms)//2]/1e9:.2f} TB/s")
