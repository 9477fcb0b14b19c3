"""Diagnostic (tooling; needs `make -C mesh-vae_amd/csrc STAMP=1`): in-kernel stamps of the vertex-patch BACKWARD kernel as
the TRAIN STEP launches it (lazy output-gradient rows, ReLU sign bytes, fused U^T pooling) -- the last instrumented launch of
a step is the patch backward (the forward's stamps of the same table are overwritten where the slots coincide).
usage: python tools/diag/stamps_step.py [--batch 64]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "mesh-vae_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("MESHVAE_LIB", os.path.join(ROOT, "mesh-vae_amd", "meshvae_hip", "libmeshvae_hip_stamp.so"))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--enc0", action="store_true", help="the first layer's kernel (k_patch_enc0) instead: the 16 -> 16 patch kernels are switched off (no_patch=2) so that its stamps are the last ones written")
args = ap.parse_args()
if args.enc0:
    os.environ["MESHVAE_DEBUG"] = "no_patch=2"
import bench
from meshvae_hip import lib
from meshvae_hip.engine import TrainStep
dev = torch.device("cuda:0")
net = bench.build_model(dev, "train5k").train()
B = args.batch
step = TrainStep(net, B, lr=1e-3, weight_decay=5e-4, use_graph=False, m_type="train", noise_seed=0, storage="f32")
x = torch.randn(B, net.num_nodes[0], 3, generator=torch.Generator().manual_seed(0))
step.x.copy_(x); step.x_gt = x.double().to(dev); step.y.copy_(torch.nn.functional.one_hot(torch.arange(B) % 2, 2))
for _ in range(5):
    step.step()
torch.cuda.synchronize()
L = lib()
rd = L.mvh_debug_read_stamps_patch
rd.restype, rd.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
assert rd(None, 1) == 0
step.step()
torch.cuda.synchronize()
buf = np.zeros(512 * 16 * 32, dtype=np.uint64)
assert rd(buf.ctypes.data, 0) == 0
t = buf.reshape(512, 16, 32).astype(np.float64)
for name, waves in ((("all waves", slice(0, 16)),) if args.enc0 else (("recurrence waves 0-7", slice(0, 8)), ("matrix waves 8-15", slice(8, 16)))):
    tt = t[:, waves, :]
    used = tt[:, :, 0] > 0
    slots = [s for s in range(32) if (tt[:, :, s][used] > 0).all()]
    print(f"== {name}: {int(used.sum())} waves stamped, slots {slots}")
    for a, b in zip(slots[:-1], slots[1:]):
        d = (tt[:, :, b] - tt[:, :, a])[used]
        print(f"  {a:2d}->{b:2d}: median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}")
    tot = (tt[:, :, slots[-1]] - tt[:, :, slots[0]])[used]
    print(f"  total {slots[0]}->{slots[-1]}: median {np.median(tot):8.0f}")
