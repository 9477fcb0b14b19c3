#!/usr/bin/env python3
"""Template mesh of the reference as a DATA fixture (build container only; see make_golden.py).

    python oracle/make_golden_template.py

Writes tests/golden/template_5k.npz = the vertices [4998,3] float64 and faces [9992,3] int64 of
/root/reference/template/template5k.obj (the reference's own data file), so that the hierarchy
generator of the build (mesh-vae_amd/mesh_operations.py) can be checked on the
GPU box / in CI against topology_5k.npz and topology_20k.npz without the reference tree.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402

REF = os.environ.get("MESHVAE_REFERENCE", "/root/reference")

if __name__ == "__main__":
    v, f = refshim.read_obj(os.path.join(REF, "template", "template5k.obj"))
    out = os.path.join(ROOT, "tests", "golden", "template_5k.npz")
    np.savez_compressed(out, verts=v, faces=f)
    print(out, v.shape, f.shape, os.path.getsize(out) // 1024, "KiB")
