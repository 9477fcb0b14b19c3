#!/usr/bin/env python3
"""Golden vectors for the input-side pre-processing, by RUNNING THE REFERENCE.

TEST INFRASTRUCTURE, build container only.  Calls the unmodified ``utils.procrustes`` of /root/reference
(utils.py:58-157, the alignment data.py:144 applies to every mesh) on seeded similarity-transformed,
noisy copies of the tiny icosphere and of a 4998-vertex procedural torus (tests/meshgen.py) -- one of them mirrored, since
orthogonal_procrustes admits reflections -- and writes inputs and outputs to tests/golden/procrustes.npz.

    python oracle/make_golden_pre.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import make_golden as mg  # noqa: E402  (installs refshim, puts the reference on sys.path)

import utils as ref_utils  # noqa: E402  (reference)


def cases(template, B, seed, mirror_at):
    g = np.random.default_rng(seed)
    out = []
    for b in range(B):
        q, _ = np.linalg.qr(g.standard_normal((3, 3)))
        if (np.linalg.det(q) < 0) != (b == mirror_at):
            q[:, 0] = -q[:, 0]
        scale = float(g.uniform(0.5, 40.0))
        shift = g.standard_normal(3) * 10.0
        noise = g.standard_normal(template.shape) * 0.01 * np.abs(template).max()
        out.append((template + noise) @ q * scale + shift)
    return np.stack(out)


def run(tag, template, pts, out):
    out[f"{tag}/template"], out[f"{tag}/pts"] = template, pts
    keys = ("mtx1", "mtx2", "disparity", "R", "s", "m")
    acc = {k: [] for k in keys}
    for p in pts:
        mtx1, mtx2, disparity, res = ref_utils.procrustes(template, p)
        for k, v in zip(keys, (mtx1, mtx2, disparity, res[0], res[1], res[2])):
            acc[k].append(np.asarray(v, dtype=np.float64))
    for k in keys:
        out[f"{tag}/{k}"] = np.stack(acc[k])
    print(tag, "disparity", out[f"{tag}/disparity"], "det R", [round(float(np.linalg.det(r)), 3) for r in out[f"{tag}/R"]])


def main():
    out = {}
    tiny = np.load(os.path.join(mg.OUT, "topology_tiny.npz"))["verts"].astype(np.float64)
    run("tiny", tiny, cases(tiny, 4, 1, mirror_at=2), out)
    # (the "5k" case is this repo's own 4998-vertex torus, tests/meshgen.py -- not the reference's template asset)
    v5k = np.load(os.path.join(mg.OUT, "hier_torus5k.npz"))["verts"].astype(np.float64)
    run("5k", v5k, cases(v5k, 2, 2, mirror_at=-1), out)
    np.savez_compressed(os.path.join(mg.OUT, "procrustes.npz"), **out)
    print("procrustes.npz", os.path.getsize(os.path.join(mg.OUT, "procrustes.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
