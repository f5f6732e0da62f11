import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "completion")); sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
import model_utils as mu
dev = "cuda:0"
ef = mu.EF_expansion(256, output_size=64, step_ratio=2, k=4).to(dev)
x = torch.randn(32, 256, 3072, device=dev, requires_grad=True)
def new():
    y = ef(x); y.square().sum().backward()
def old():
    edge_in = mu.get_graph_feature(x, 4, minus_center=False).permute(0, 1, 3, 2).contiguous()
    edge = F.relu(torch.cat((ef.conv1(edge_in), edge_in), 1))
    edge = F.relu(ef.conv2(edge))
    edge = edge.permute(0, 2, 3, 1).contiguous().view(32, 4, 3072 * 2, 64).permute(0, 3, 1, 2)
    y = ef.conv3(edge).max(dim=2)[0]; y.square().sum().backward()
for name, f in (("restructured", new), ("edge tensor", old)):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); print("EF_expansion (32,256,3072) k=4 step 2 fwd+bwd, %s: %.2f ms" % (name, (time.perf_counter() - t0) / 5 * 1e3))
