"""Diagnostic (tooling): run the reference-API loop (zero_grad / net() / backward / Adam.step) for a few steps so that
`rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/diag/ref_loop_trace.py` records its kernels;
tools/diag/ref_loop_gaps.py reads the trace and prints busy time and the idle gaps per step.
env: REF_ASSIGN=1 (net.grad_mode="assign"), REF_FUSED=1 (Adam fused=True), REF_STEPS (default 40)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
dev = torch.device("cuda:0")
B = 64
x = torch.randn(B, 4998, 3, generator=torch.Generator().manual_seed(0)).to(dev)
x_gt = x.double()
y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
d = bench.RefBatch(x) if hasattr(bench, "RefBatch") else type("D", (), dict(x=x.reshape(-1, 3), num_graphs=B, edge_index=None))()
net = bench.build_model(dev).train()
if os.environ.get("REF_ASSIGN") == "1":
    net.grad_mode = "assign"
opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-4, **({"fused": True} if os.environ.get("REF_FUSED") == "1" else {}))
n = int(os.environ.get("REF_STEPS", "40"))
for i in range(100 + n):
    if i == 100:
        torch.cuda.synchronize()
        torch.zeros(7, device=dev).fill_(1.0)          # marker: the traced steps start after this fill
        torch.cuda.synchronize()
    opt.zero_grad()
    loss = net(d, x_gt, y, m_type="train")[0]
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("done", float(loss))
