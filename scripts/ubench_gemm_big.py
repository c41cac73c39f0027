import os, sys
sys.path.insert(0, "/root/repo")
import torch
from spatialrgpt_amd import ops
dev="cuda"
for name,M,N,K in [("sq4096",4096,4096,4096),("sq8192",8192,8192,8192),("vit b8 fc1",11664,4304,1152),("prefill b8 gate/up",2072,28672,4096),("prefill b8 down",2072,4096,14336)]:
    Ws=[torch.randn((N,K),device=dev,dtype=torch.bfloat16)*0.02 for _ in range(2)]
    a=torch.randn((M,K),device=dev,dtype=torch.bfloat16); out=torch.empty((M,N),device=dev,dtype=torch.bfloat16)
    for W in Ws: ops.gemm(a,W,out=out)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        for W in Ws: ops.gemm(a,W,out=out)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/10
    print(f"{name:20s} {us:9.1f} us {2*M*N*K/us/1e6:8.1f} TF/s")
