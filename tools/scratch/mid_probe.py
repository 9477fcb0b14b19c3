import os, sys, ctypes, torch
sys.path.insert(0, "/root/repo/mesh-vae_amd"); sys.path.insert(0, "/root/repo/tests")
from conftest import CFG_5K, ROOT
from meshvae_hip import lib
from meshvae_hip.engine import NativeStep
from model import load_topology
from models.cheb_VAE import cheb_VAE
dev = torch.device("cuda:0")
D, U, A, nn_ = load_topology(os.path.join(ROOT, "tests", "golden", "topology_5k.npz"), dev)
net = cheb_VAE(3, dict(CFG_5K, dropout=0.0), D, U, A, nn_, model="optimal_sigma_VAE").to(dev).train()
from meshvae_hip import debug_switch
import itertools
for B, exp in itertools.product((64,), (0, 1, 2, 4)):
  with debug_switch('l0_wide', exp):
      x = torch.randn(B, 4998, 3, device=dev); y = torch.nn.functional.one_hot(torch.arange(B) % 2, 2).to(dev)
      nat = NativeStep(net, B)
      t = torch.zeros(64, dtype=torch.int64, device=dev)
      L = ctypes.CDLL(lib()._name) if hasattr(lib(), "_name") else lib()
      L.mvh_debug_mid_tlog.argtypes = [ctypes.c_void_p]; L.mvh_debug_mid_tlog.restype = None
      L.mvh_debug_mid_tlog(t.data_ptr())
      for _ in range(3):
          nat.forward_backward(x, x, y, eps=None, backward=False); torch.cuda.synchronize()
      tt = t.cpu().tolist()
      print("exp", exp, "shader MHz", round((tt[62]-tt[56])/((tt[6]-tt[0])/100.0)), "B", B, "marks (us, 100 MHz clock):", [round((tt[i + 1] - tt[i]) / 100.0, 2) for i in range(6)], "total", (tt[6] - tt[0]) / 100.0)
      for nm, base, start in (("conv0", 8, tt[2]), ("conv1", 32, tt[4])):
          prev = start; out = []
          for k in range(5, -1, -1):
              out.append([round((tt[base + 3 * k + j] - (prev if j == 0 else tt[base + 3 * k + j - 1])) / 100.0, 2) for j in range(3)])
              prev = tt[base + 3 * k + 2]
          print("  ", nm, "per order k=5..0 [mfma, phase1+barrier, gather+barrier]:", out)
      L.mvh_debug_mid_tlog(None)
