import os, sys
sys.path.insert(0, '/root/repo')
import torch, torch.nn.functional as F
from mvp_benchmark_amd.pointwise import mfma_linear
dev='cuda:0'
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (B,cin,cout,L) in [(32,1029,512,16384),(32,1028,512,16384),(32,1024,512,16384),(32,512,512,16384),(32,512,3,16384),(32,128,256,2048),(32,256,512,2048),(32,512,1024,2048),(32,3,128,2048)]:
    x=torch.randn(B,cin,L,device=dev); w=torch.randn(cout,cin,device=dev); b=torch.randn(cout,device=dev); w3=w.unsqueeze(2).contiguous()
    t1=timeit(lambda: mfma_linear(x,w,b,relu=True)); t2=timeit(lambda: torch.relu_(F.conv1d(x,w3,b)))
    fl=2.0*B*cin*cout*L
    print((B,cin,cout,L), "mfma %.3f ms (%.1f TF)  lib %.3f ms (%.1f TF)"%(t1,fl/t1/1e9,t2,fl/t2/1e9), flush=True)
