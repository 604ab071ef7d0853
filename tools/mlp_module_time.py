import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, ROOT+"/tiny-nerf-pytorch_amd"); sys.path.insert(0, ROOT+"/tiny-nerf-pytorch_amd/src")
import torch, numpy as np
import nerf
from tnerf import ops, lib
dev=torch.device("cuda:0")
for flag in ("", "mfma32"):
    os.environ["TNERF_FP32_PIPE"]=flag
    for (L,hid,depth,skip,M) in ((6,256,8,4,262144),(10,128,4,2,131072)):
        torch.manual_seed(0)
        m=nerf.TinyNeRF(6*L+3,hid,depth,skip).to(dev)
        x=torch.randn(M,6*L+3,device=dev)
        def fb():
            rgb,sig=m(x); (rgb.sum()+sig.sum()).backward()
        def f():
            with torch.no_grad(): m(x)
        out=[]
        for name,fn in (("fwd(no grad)",f),("fwd+bwd",fb)):
            for _ in range(5): fn()
            torch.cuda.synchronize(); t0=time.perf_counter()
            for _ in range(20): fn()
            torch.cuda.synchronize(); out.append(f"{name} {(time.perf_counter()-t0)/20*1e3:.3f} ms")
        print(f"pipe={flag or 'x3'} {depth}x{hid} M={M}: "+"  ".join(out), flush=True)
