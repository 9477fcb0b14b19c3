import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, ctypes
import bench
from meshvae_hip.engine import NativeStep
dev = torch.device("cuda:0")
net = bench.build_model(dev).train()
nat = NativeStep(net, 4)
d = nat.desc
print("down0.patch", d.down[0].patch, "flags", d.down[0].flags, "n_rows", d.down[0].n_rows, "n_cols", d.down[0].n_cols, "rowptr", d.down[0].rowptr)
from meshvae_hip import PatchPlanStruct
if d.down[0].patch:
    pl = ctypes.cast(d.down[0].patch, ctypes.POINTER(PatchPlanStruct)).contents
    print({f[0]: getattr(pl, f[0]) for f in PatchPlanStruct._fields_})
x = torch.randn(4, 4998, 3, device=dev); y = torch.nn.functional.one_hot(torch.arange(4) % 2, 2).to(dev)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    nat.forward_backward(x, x, y, eps=torch.randn(4, net.z, device=dev), drop_u=None)
    torch.cuda.synchronize()
names = sorted({e.name for e in prof.events() if "mvh" in e.name})
print("\n".join(n[:90] for n in names))
