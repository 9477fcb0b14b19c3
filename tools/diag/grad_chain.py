"""Diagnostic (tooling): the activation gradients of the native step's decoder chain against the float64 oracle, tensor
by tensor and mesh by mesh -- find the first tensor that is wrong at B = 64 and where."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "mesh-vae_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import torch.nn.functional as F
import test_gpu_b64 as T
from conftest import CFG_5K
from oracle import cheb_oracle as O

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pdrop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
T.P_DROP = pdrop
net = T._build(CFG_5K, "topology_5k.npz", dev, dropout=pdrop).train()
x, y, eps, g = T._inputs(net, B)
H, flat = net.num_hidden, net.dec_lin_2.out_features
drop_u = torch.rand(B * (3 * H + flat), generator=g)
nat, got = T._native(net, B, x, y, eps, drop_u)
if pdrop == 0.0:
    drop_u = torch.ones_like(drop_u)


class Rec(O.OracleVAE):
    def decoder(self, z):
        self.t = {}
        x = self._drop(F.relu(F.linear(z, self.p["dec_lin.weight"], self.p["dec_lin.bias"])))
        self.t["d1"] = x
        x = self._drop(F.relu(F.linear(x, self.p["dec_lin_2.weight"], self.p["dec_lin_2.bias"])))
        self.t["d2"] = x
        x = x.reshape(x.shape[0], -1, self.filters[-1])
        for i in range(self.n_layers):
            x = O.surface_pool(x, *self.topo.U[-i - 1])
            self.t[f"decU{i}"] = x
            pre = self._conv(f"cheb_dec.{i}", x, self.n_layers - i - 1)
            self.t[f"pre{i}"] = pre
            x = F.relu(pre)
            self.t[f"decC{i}"] = x
        for v in self.t.values():
            v.retain_grad()
        return self._conv(f"cheb_dec.{self.n_layers}", x, len(self.edge) - 1)


dt = torch.float64
ora = Rec(dict(CFG_5K, dropout=pdrop), O.Topology(np.load(os.path.join(ROOT, "tests", "golden", "topology_5k.npz"))),
          {k: v.cpu() for k, v in net.state_dict().items()}, requires_grad=True, dtype=dt)
ora.training = True
blocks = T._drop_blocks(drop_u, B, H, flat)
ora._drop = lambda t: (lambda u: torch.where(u >= max(pdrop, 1e-30), t / (1.0 - pdrop), torch.zeros_like(t)))(blocks.pop(0))
out = ora.forward(x.to(dt), x.to(dt), y.to(dt), "train", eps=eps.to(dt))
out[0].backward()
n = net.n_layers
print(f"B={B} p={pdrop}")
for name, idx, key in [("g_decC", 3, "decC3"), ("g_decC", 2, "decC2"), ("g_decC", 1, "decC1"), ("g_decC", 0, "decC0"), ("g_d2", 0, "d2"), ("g_d1", 0, "d1"),
                       ("decC", 3, "decC3"), ("decU", 3, "decU3"), ("decC", 2, "decC2")]:
    want = ora.t[key].grad if name.startswith("g_") else ora.t[key].detach()
    have = nat.ws_tensor(name, idx).cpu().double().reshape(want.shape)
    err = (have - want).flatten(1)
    per_mesh = err.norm(dim=1) / want.flatten(1).norm(dim=1)
    worst = int(per_mesh.argmax())
    nbad = int((per_mesh > 1e-5).sum())
    msg = f"{name}[{idx}] shape {tuple(want.shape)} rel {float(err.norm() / want.norm()):.2e}; meshes with rel > 1e-5: {nbad}; worst mesh {worst} rel {float(per_mesh[worst]):.2e}"
    if nbad:
        e = (have[worst] - want[worst]).abs()
        rows = (e.reshape(e.shape[0], -1).max(dim=1).values > 1e-6 * float(want[worst].abs().max())).nonzero().flatten()
        msg += f"; bad meshes {[int(i) for i in (per_mesh > 1e-5).nonzero().flatten()[:16]]}; bad rows in worst mesh: {len(rows)} first {rows[:12].tolist()} last {rows[-4:].tolist()}"
    print(msg, flush=True)

# ReLU sign disagreements between this library's forward and the float64 oracle, with the oracle's pre-activation there
for i in range(n):
    have = nat.ws_tensor("decC", i).cpu().reshape(ora.t[f"decC{i}"].shape)
    pre = ora.t[f"pre{i}"].detach()
    diff = ((have > 0) != (pre > 0)).nonzero()
    print(f"decoder stage {i}: {len(diff)} sign disagreements of {pre.numel()}; " +
          "; ".join(f"mesh {int(b)} vertex {int(v)} ch {int(c)}: oracle pre {float(pre[b, v, c]):.3e} ours {float(have[b, v, c]):.3e}" for b, v, c in diff[:6]))
