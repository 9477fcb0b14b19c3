"""Debug aid: which part of the micro-batched step breaks hipGraph capture?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd"))
import torch
from bench import build_model
from meshvae_hip.engine import NativeStep
mode = sys.argv[1]
dev = torch.device("cuda:0")
net = build_model(dev); net.train(); net._prepare()
B = 32
x = torch.randn(B, 4998, 3, device=dev); y = torch.nn.functional.one_hot(torch.arange(B, device=dev) % 2, 2)
eps = torch.randn(B, 16, device=dev)
s1, s2, s3 = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for p in net.parameters():
    p.grad = torch.zeros_like(p)
natA = NativeStep(net, B, side_stream=s2)
natB = NativeStep(net, B, grads=[torch.zeros_like(p) for p in net.parameters()], side_stream=s3)
bwd = "fwdonly" not in mode
def chain(nat, u=None):
    nat.forward_backward(x, x, y, eps, u, backward=bwd)
side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    chain(natA); chain(natB)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream(dev)
    if "A" in mode:
        chain(natA)
    if "B" in mode:
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            chain(natB)
        cur.wait_stream(s1)
g.replay(); torch.cuda.synchronize()
print(mode, "ok")
