"""Diagnostic (tooling; needs `make -C mesh-vae_amd/csrc STAMP=1`): phase shares of the LDS-resident conv kernel from the
in-kernel stamps of the diagnostic build.  usage: MESHVAE_LIB=.../libmeshvae_hip_stamp.so python tools/diag/stamps_lds.py [--bwd] [--cin 16 --cout 16 --k 6]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "mesh-vae_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("MESHVAE_LIB", os.path.join(ROOT, "mesh-vae_amd", "meshvae_hip", "libmeshvae_hip_stamp.so"))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=16); ap.add_argument("--cout", type=int, default=16)
ap.add_argument("--k", type=int, default=6); ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--level", type=int, default=0); ap.add_argument("--bwd", action="store_true"); ap.add_argument("--dw", action="store_true")
ap.add_argument("--patch", action="store_true", help="the vertex-patch kernels (cheb_patch.hip): --bwd = dX + dW in one launch")
args = ap.parse_args()
from meshvae_hip import check, lib, topology
from meshvae_hip.functional import workspace
from nn.conv import ChebConv_batch
dev = torch.device("cuda:0")
z = np.load(os.path.join(ROOT, "tests", "golden", "topology_5k.npz"))
N = int(z["num_nodes"][args.level])
ei = torch.from_numpy(np.vstack([z[f"A{args.level}_row"], z[f"A{args.level}_col"]]).astype(np.int64)).to(dev)
ei, nrm = ChebConv_batch.norm(ei, N)
op = topology.laplacian(ei, nrm, N)
L = lib()
B, Cin, Cout, K = args.batch, args.cin, args.cout, args.k
x = torch.randn(B, N, Cin, device=dev); W = torch.randn(K, Cin, Cout, device=dev) * 0.1; bias = torch.randn(Cout, device=dev) * 0.1
out = torch.empty(B, N, Cout, device=dev); dout = torch.randn(B, N, Cout, device=dev)
dx = torch.empty_like(x)
signs = torch.empty(B, N, max(Cout // 4, 1), dtype=torch.uint8, device=dev)
wsb = max(L.mvh_cheb_conv_ws_bytes(B, N, Cin, Cout, K), L.mvh_cheb_conv_bwd_ws_bytes(B, N, Cin, Cout, K))
ws = workspace(wsb, dev)
st = torch.cuda.current_stream(dev).cuda_stream
dW, db = torch.empty_like(W), torch.empty_like(bias)
rd = L.mvh_debug_read_stamps_patch if args.patch else (L.mvh_debug_read_stamps_dw if args.dw else L.mvh_debug_read_stamps_lds)
rd.restype, rd.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]

def run():
    if args.dw:
        check(L.mvh_cheb_conv_bwd_signs(st, op.fwd.ref, op.bwd.ref, x.data_ptr(), W.data_ptr(), out.data_ptr(), signs.data_ptr(),
                                        dout.data_ptr(), None, dW.data_ptr(), db.data_ptr(), B, N, Cin, Cout, K, ws.data_ptr(), wsb))
    elif args.bwd:
        check(L.mvh_cheb_conv_bwd_signs(st, op.fwd.ref, op.bwd.ref, x.data_ptr(), W.data_ptr(), out.data_ptr(), signs.data_ptr(),
                                        dout.data_ptr(), dx.data_ptr(), dW.data_ptr() if args.patch else None, db.data_ptr() if args.patch else None, B, N, Cin, Cout, K, ws.data_ptr(), wsb))
    else:
        check(L.mvh_cheb_conv_fwd_signs(st, op.fwd.ref, x.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), signs.data_ptr(),
                                        B, N, Cin, Cout, K, ws.data_ptr(), wsb))
check(L.mvh_cheb_conv_fwd_signs(st, op.fwd.ref, x.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), signs.data_ptr(), B, N, Cin, Cout, K, ws.data_ptr(), wsb))
for _ in range(3):
    run()
torch.cuda.synchronize()
assert rd(None, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); e1.synchronize()
buf = np.zeros(512 * 16 * 32, dtype=np.uint64)
assert rd(buf.ctypes.data, 0) == 0
t = buf.reshape(512, 16, 32).astype(np.float64)
used = t[:, :, 0] > 0
t0 = t[:, :, 0][used].min()
span = t[used].max() - t0
print(f"{'dW' if args.dw else 'dX' if args.bwd else 'fwd'} N={N} {Cin}->{Cout} K={K} B={B}: event time {e0.elapsed_time(e1) * 1e3:.1f} us (diagnostic build), stamp span {span:.0f} ticks, "
      f"{int(used.sum())} waves stamped")
slots = [s for s in range(32) if (t[:, :, s][used] > 0).all()]
print("slots present:", slots)
rel = {s: t[:, :, s][used] - t0 for s in slots}
print("slot: median arrival / min / max (ticks since the first wave's start); delta of medians")
prev = None
for s in slots:
    med = np.median(rel[s])
    print(f"  {s:2d}: {med:9.0f} {rel[s].min():9.0f} {rel[s].max():9.0f}   +{(med - prev) if prev is not None else 0:8.0f}")
    prev = med
# per-wave segment lengths (median over waves)
print("per-wave segment medians:")
for a, b in zip(slots[:-1], slots[1:]):
    d = (t[:, :, b] - t[:, :, a])[used]
    print(f"  {a:2d}->{b:2d}: median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}")
