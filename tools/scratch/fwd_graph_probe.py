"""Is a hipGraph of the FORWARD alone (a pure chain of ~17 kernels, no lanes) faster than its eager launches at B = 64?
forward_backward(backward=False) eager against the same captured into a torch.cuda.CUDAGraph; and the whole step with a
graphed forward + eager backward (timing only: the replayed forward does not tell the backward about its stack)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd"))
import torch, bench
from meshvae_hip.engine import NativeStep
dev = torch.device("cuda:0")
B = 64
net = bench.build_model(dev).train()
nat = NativeStep(net, B)
x = torch.randn(B, 4998, 3, device=dev); xg = x.double(); y = torch.nn.functional.one_hot(torch.arange(B, device=dev) % 2, 2).float()
eps = torch.randn(B, net.z, device=dev); du = torch.rand(B * nat.u_cols, device=dev)
def t(fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
fwd = lambda: nat.forward_backward(x, xg, y, eps=eps, drop_u=du, backward=False)
print(f"forward eager: {t(fwd):.1f} us")
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    for _ in range(3): fwd()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fwd()
print(f"forward graph replay: {t(g.replay):.1f} us")
both = lambda: nat.forward_backward(x, xg, y, eps=eps, drop_u=du)
print(f"forward + backward eager: {t(both):.1f} us")
import ctypes
from meshvae_hip import lib, check
L = lib()
def bwd_only():
    d = ctypes.byref(nat.desc); st = torch.cuda.current_stream(dev).cuda_stream
    check(L.mvh_vae_backward(st, d, nat._P, nat._G, x.data_ptr(), y.data_ptr(), xg.data_ptr(), 1, eps.data_ptr(), du.data_ptr(), B,
                             nat.log_sigma, None, nat.recon.data_ptr(), nat.y_hat.data_ptr(), nat.mu.data_ptr(), nat.logvar.data_ptr(),
                             nat.ws.data_ptr(), nat.ws_bytes, None))
def mixed():
    g.replay(); bwd_only()
print(f"graphed forward + eager backward (timing only): {t(mixed):.1f} us")
