"""cProfile of the reference loop's calling thread (module path, launcher on): where net() and backward() spend host time"""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, bench
dev = torch.device("cuda:0")
B = 64
x = torch.randn(B, 4998, 3, generator=torch.Generator().manual_seed(0)).to(dev)
x_gt = x.double()
y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
d = bench.RefBatch(x) if hasattr(bench, "RefBatch") else type("D", (), dict(x=x.reshape(-1, 3), num_graphs=B, edge_index=None))()
net = bench.build_model(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-4)
def step():
    opt.zero_grad(); loss = net(d, x_gt, y, m_type="train")[0]; loss.backward(); opt.step()
for _ in range(300): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(500): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
