"""conv -> max over the positions, forward: the GEMM followed by a max kernel against the max inside the GEMM's epilogue
(mvp_pointwise_mfma_max).  python tools/bench_conv_max_fused.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvp_benchmark_amd import pointwise as pw
dev="cuda:0"
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
for (B,cin,cout,L) in ((64,512,1024,2048),(32,512,1024,2048),(64,512,1024,384),(32,1800,1024,64)):
    x=torch.randn(B,cin,L,device=dev); w=torch.randn(cout,cin,1,device=dev)/cin**0.5; b=torch.randn(cout,device=dev)
    with torch.no_grad():
        t_un=timeit(lambda: pw.pointwise_conv(x,w,b).flatten(2).max(dim=2))
        t_f=timeit(lambda: pw.mfma_conv_max(x,w,b))
        t_g=timeit(lambda: pw.pointwise_conv(x,w,b))
    print("(%d,%d->%d,%d): GEMM alone %.3f ms, GEMM + max kernel %.3f ms, fused epilogue %.3f ms" % (B,cin,cout,L,t_g,t_un,t_f if pw.mfma_conv_max(x,w,b) is not None else float('nan')))
