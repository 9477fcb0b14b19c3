"""Procedural test meshes (this repo's own data; nothing here comes from the reference's assets).

torus_mesh(51, 98) has exactly the 5k template's counts -- 4998 vertices, 9996 triangles, genus 1 -- with an
irregular, anisotropic surface (two superposed bumps and a sheared parametrisation) so that quadric collapse
costs are distinct and the decimator's heap order is exercised like on a scanned surface.
"""
import numpy as np


def torus_mesh(nu=51, nv=98, R=1.0, r=0.37):
    """-> verts [nu*nv, 3] float64, faces [2*nu*nv, 3] int64 (consistent orientation, closed, genus 1)."""
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    u = 2.0 * np.pi * (i + 0.31 * np.sin(2.0 * np.pi * j / nv)) / nu        # around the tube
    v = 2.0 * np.pi * (j + 0.23 * np.cos(2.0 * np.pi * i / nu)) / nv        # around the axis
    rr = r * (1.0 + 0.18 * np.sin(3.0 * v + 0.4) * np.cos(2.0 * u) + 0.07 * np.cos(5.0 * v - u))
    RR = R * (1.0 + 0.12 * np.sin(2.0 * v + 1.1))
    x = (RR + rr * np.cos(u)) * np.cos(v)
    y = (RR + rr * np.cos(u)) * np.sin(v) * 0.83
    z = rr * np.sin(u) * (1.0 + 0.25 * np.cos(v))
    verts = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float64)
    idx = np.arange(nu * nv).reshape(nu, nv)
    a, b = idx, np.roll(idx, -1, axis=0)
    c, d = np.roll(idx, -1, axis=1), np.roll(np.roll(idx, -1, axis=0), -1, axis=1)
    faces = np.concatenate([np.stack([a, b, d], -1).reshape(-1, 3), np.stack([a, d, c], -1).reshape(-1, 3)])
    return verts, faces.astype(np.int64)


def subdivide(v, f):
    """1 -> 4 midpoint subdivision (BASELINE configs[3]'s way of making a hi-res template)."""
    v = [p for p in v]
    cache, nf = {}, []

    def mid(a, b):
        key = (min(a, b), max(a, b))
        if key not in cache:
            v.append(0.5 * (v[a] + v[b]))
            cache[key] = len(v) - 1
        return cache[key]

    for a, b, c in f:
        ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
        nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
    return np.stack(v), np.asarray(nf, dtype=np.int64)
