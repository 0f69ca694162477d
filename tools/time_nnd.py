import sys,os
sys.path.insert(0,os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import torch
from genre_shapehd_amd.toolbox.nndistance._ext import my_lib
dev=torch.device("cuda:0")
for B in (1,2,3,4,8,16,32,64,128):
    n=2048
    a=torch.rand((B,n,3),device=dev); b=torch.rand((B,n,3),device=dev)
    d1=torch.empty((B,n),device=dev); d2=torch.empty_like(d1); i1=torch.empty((B,n),device=dev,dtype=torch.int32); i2=torch.empty_like(i1)
    g=torch.cuda.CUDAGraph()
    s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): my_lib.nnd_forward_cuda(a,b,d1,d2,i1,i2)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(10): my_lib.nnd_forward_cuda(a,b,d1,d2,i1,i2)
    g.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/50
    print("B=%d  %.1f us  %.1f TFLOP/s"%(B,us,2*B*n*n*8/us/1e6))
