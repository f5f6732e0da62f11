import sys, os, torch
sys.path.insert(0, '/root/repo')
from mvp_benchmark_amd import _lib
if os.environ.get('MVP_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['MVP_LIB'])
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (b, n, m, k) in [(64, 3072, 3072, 10), (64, 3072, 3072, 20), (64, 3072, 1536, 16), (64, 1536, 1536, 20), (64, 2048, 2048, 16), (64, 768, 768, 20), (32, 2048, 2048, 16)]:
    x = torch.rand(b, n, 3, generator=g).to(dev); c = x[:, :m].contiguous()
    res = {}
    for name, fn in (("sorted", "mvp_knn_sorted"), ("exhaustive", "mvp_knn")):
        nbytes = _lib.knn_scratch_bytes(b, n, m)
        scratch = torch.zeros(max(nbytes,1), dtype=torch.uint8, device=dev)
        idx = torch.zeros(b, m, k, dtype=torch.int32, device=dev); d2 = torch.zeros(b, m, k, device=dev)
        def run():
            if fn == "mvp_knn_sorted":
                _lib.call(fn, dev, b, n, m, k, x, c, idx, d2, scratch, nbytes)
            else:
                _lib.call(fn, dev, b, n, m, k, x, c, idx, d2)
        try:
            run(); torch.cuda.synchronize()
        except Exception as e:
            print(name, (b,n,m,k), "failed:", e); continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / 5, idx.clone())
    same = torch.equal(res["sorted"][1], res["exhaustive"][1]) if len(res) == 2 else None
    print((b,n,m,k), {k_: round(v[0],3) for k_, v in res.items()}, "identical", same, flush=True)
