import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_patch as T
a = T._step_5k({})
b = T._step_5k({"no_enc0_patch": 1})
print("loss", a[0], b[0])
print("recon maxdiff", float((a[1] - b[1]).abs().max()), "z", float((a[2] - b[2]).abs().max()))
for k in a[3]:
    print(k, float((a[3][k] - b[3][k]).abs().max()), float(b[3][k].abs().max()))
