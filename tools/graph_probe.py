"""Probe: does multi-stream hipGraph capture through torch work on this box? (debug aid)"""
import sys
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
dev = torch.device("cuda:0")
a = torch.zeros(1000, device=dev); b = torch.zeros(1000, device=dev)
s1 = torch.cuda.Stream(dev)
# warm-up
b.add_(1); a.add_(1); torch.rand(10, device=dev); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream(dev)
    s1.wait_stream(cur)
    with torch.cuda.stream(s1):
        if mode == "rand":
            u = torch.rand(1000, device=dev)
            b.add_(u)
        elif mode == "memset":
            b.zero_()
        else:
            b.add_(1)
    a.add_(1)
    cur.wait_stream(s1)
    a.add_(b)
g.replay(); torch.cuda.synchronize()
print(mode, "ok", float(a[0]), float(b[0]))
