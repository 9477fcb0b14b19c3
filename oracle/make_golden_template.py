#!/usr/bin/env python3
"""Hierarchy fixture for the build's own decimator on a 5k-vertex mesh, by RUNNING THE REFERENCE.

TEST INFRASTRUCTURE, build container only (see make_golden.py).  The reference's template OBJ is a
third-party asset without a licence, so its geometry is NOT committed: the committed 5k-vertex case is
this repo's own procedural torus (tests/meshgen.py: 4998 vertices, 9996 faces, genus 1 like the
template) pushed through the reference's unmodified mesh_operations.generate_transform_matrices
(mesh_operations.py:253-278, factors 4,4,4,4).  tests/test_mesh_operations.py additionally checks the real
template against topology_5k.npz / topology_20k.npz wherever /root/reference is readable.

    python oracle/make_golden_template.py     # ~25 s -> tests/golden/hier_torus5k.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_golden as G  # noqa: E402  (installs the stand-ins and imports the reference modules)
import meshgen  # noqa: E402

if __name__ == "__main__":
    v, f = meshgen.torus_mesh(51, 98)
    M, A, D, U = G.hierarchy(v, f, [4, 4, 4, 4])
    topo = G.pack_topology(M, A, D, U)
    topo["verts"], topo["faces"] = v, f
    out = os.path.join(G.OUT, "hier_torus5k.npz")
    np.savez_compressed(out, **topo)
    print(out, [len(m.v) for m in M], os.path.getsize(out) // 1024, "KiB")
