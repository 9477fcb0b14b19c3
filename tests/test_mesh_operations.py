"""CPU: the build's hierarchy generator (mesh-vae_amd/mesh_operations.py) against the
A / D / U hierarchies captured from the reference's own generator (tests/golden/topology_*.npz):
adjacency and decimation exactly (entry order included), upsampling weights to 1e-12 (its closest-point
search is the stand-in's in both cases, psbody being unpinned -- see DESIGN.md section 2)."""
import os
import time

import numpy as np
import pytest

import mesh_operations as mo
from conftest import load_golden
from meshgen import subdivide, torus_mesh

# The reference's template OBJ is a third-party asset and is not committed: the real-template cases run
# wherever the reference tree is readable (the build container) and skip elsewhere; the committed 5k-vertex
# case is this repo's own torus (hier_torus5k.npz, written by oracle/make_golden_template.py from the
# reference's generator).
REF_TEMPLATE = os.path.join(os.environ.get("MESHVAE_REFERENCE", "/root/reference"), "template", "template5k.obj")
needs_ref_template = pytest.mark.skipif(not os.path.exists(REF_TEMPLATE), reason="reference template OBJ not present")


def _check(M, A, D, U, topo, n_levels):
    assert [len(m.v) for m in M] == [int(x) for x in topo["num_nodes"][:n_levels]]
    for i in range(n_levels):
        assert np.array_equal(A[i].row, topo[f"A{i}_row"]) and np.array_equal(A[i].col, topo[f"A{i}_col"]), f"A{i}"
        assert np.array_equal(A[i].data.astype(np.float32), topo[f"A{i}_val"]), f"A{i} values"
    for i in range(n_levels - 1):
        assert np.array_equal(D[i].row, topo[f"D{i}_row"]) and np.array_equal(D[i].col, topo[f"D{i}_col"]), f"D{i}"
        assert tuple(D[i].shape) == tuple(int(x) for x in topo[f"D{i}_shape"])
        assert np.array_equal(U[i].row, topo[f"U{i}_row"]) and np.array_equal(U[i].col, topo[f"U{i}_col"]), f"U{i}"
        np.testing.assert_allclose(U[i].data.astype(np.float32), topo[f"U{i}_val"], rtol=0, atol=1e-6, err_msg=f"U{i}")


def test_tiny_icosphere_hierarchy_matches_reference(topotiny_npz):
    mesh = mo.Mesh(v=topotiny_npz["verts"], f=topotiny_npz["faces"])
    M, A, D, U = mo.generate_transform_matrices(mesh, [4, 4])
    _check(M, A, D, U, topotiny_npz, 3)


def test_torus_5k_hierarchy_matches_reference():
    """4998 vertices / 9996 faces / genus 1 (the template's counts) on this repo's own geometry."""
    topo = load_golden("hier_torus5k.npz")
    v, f = torus_mesh(51, 98)
    np.testing.assert_allclose(v, topo["verts"], rtol=0, atol=1e-15)   # the generator is reproducible
    assert np.array_equal(f, topo["faces"])
    M, A, D, U = mo.generate_transform_matrices(mo.Mesh(v=topo["verts"], f=topo["faces"]), [4, 4, 4, 4])
    _check(M, A, D, U, topo, 5)


@needs_ref_template
def test_template_5k_hierarchy_matches_reference(topo5k_npz):
    mesh = mo.Mesh(filename=REF_TEMPLATE)
    t0 = time.time()
    M, A, D, U = mo.generate_transform_matrices(mesh, [4, 4, 4, 4])
    print(f"5k hierarchy in {time.time() - t0:.1f} s")
    _check(M, A, D, U, topo5k_npz, 5)


def test_obj_reader_and_edges(tmp_path):
    p = tmp_path / "t.obj"
    p.write_text("# c\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1\nf 2 4 3\n")
    m = mo.Mesh(filename=str(p))
    assert m.v.shape == (4, 3) and m.f.tolist() == [[0, 1, 2], [1, 3, 2]]
    e = mo.get_vertices_per_edge(m.v, m.f)
    assert sorted(map(tuple, e.tolist())) == [(0, 1), (0, 2), (1, 2), (1, 3), (2, 3)]
    with pytest.raises(Exception, match="factor or n_verts_desired"):
        mo.qslim_decimator_transformer(m)


@needs_ref_template
def test_subdivided_20k_hierarchy_matches_reference(topo20k_npz):
    """BASELINE configs[3]'s template: coplanar sub-faces give exactly tied / zero collapse costs, so this is
    the case where the heap's tie-breaking and the last bits of the quadrics decide the result (~2 min)."""
    t = mo.Mesh(filename=REF_TEMPLATE)
    v, f = subdivide(t.v, t.f)
    M, A, D, U = mo.generate_transform_matrices(mo.Mesh(v=v, f=f), [4, 4, 4, 4, 4])
    _check(M, A, D, U, topo20k_npz, 6)


def test_get_model_builds_the_hierarchy_from_the_template(tmp_path, topotiny_npz, capsys):
    """model.get_model with the reference's own config keys (`template`, `downsampling_factors`): same
    topology tensors and the same seeded weights as the fixture-driven construction."""
    import torch
    from conftest import TINY_CFG
    from model import get_model, load_topology
    obj = tmp_path / "tiny.obj"
    with open(obj, "w") as fp:
        for p in topotiny_npz["verts"]:
            fp.write("v %.17g %.17g %.17g\n" % tuple(p))
        for a, b, c in topotiny_npz["faces"]:
            fp.write(f"f {a + 1} {b + 1} {c + 1}\n")
    cfg = dict(TINY_CFG, template=str(obj), downsampling_factors=[4, 4], type="cheb_VAE", model="optimal_sigma_VAE",
               checkpoint_dir=str(tmp_path))
    torch.manual_seed(666)
    net = get_model(cfg, "cpu")
    capsys.readouterr()
    D_t, U_t, A_t, nn_ = load_topology(str(load_golden.__globals__["GOLDEN"]) + "/topology_tiny.npz", "cpu")
    assert net.num_nodes == nn_
    for a, b in zip(net.adjacency_matrices, A_t):
        assert torch.equal(a._indices(), b._indices())
    for a, b in zip(net.downsample_matrices, D_t):
        assert torch.equal(a._indices(), b._indices()) and torch.equal(a._values(), b._values())
    for a, b in zip(net.upsample_matrices, U_t):
        assert torch.equal(a._indices(), b._indices()) and torch.equal(a._values(), b._values())
    assert (tmp_path / "initial_weight.pt").exists()


def test_pruned_closest_point_equals_exhaustive_scan():
    """nearest_on_surface's candidate pruning returns exactly what the all-triangles scan returns: faces, region
    codes and hit points, for points on, near and far from the surface (ties resolve to the lowest face)."""
    mesh = mo.Mesh(*torus_mesh(51, 98))
    g = np.random.default_rng(0)
    v = mesh.v
    pick = g.choice(len(v), 120, replace=False)
    span = np.ptp(v, axis=0).max()
    pts = np.concatenate([v[pick[:40]],                                            # exactly on vertices (ties)
                          v[pick[40:80]] + g.standard_normal((40, 3)) * 0.01 * span,  # near the surface
                          v[pick[80:]] + g.standard_normal((40, 3)) * 2.0 * span])    # far away
    sf = np.asarray(mesh.f, dtype=np.int64)
    a, b, c = v[sf[:, 0]], v[sf[:, 1]], v[sf[:, 2]]
    f0, r0, h0 = mo._nearest_exhaustive(pts, a, b - a, c - a, b, c)
    f1, r1, h1 = mo.nearest_on_surface(mesh, pts)
    f2, r2, h2 = mo.nearest_on_surface(mesh, pts, chunk_pairs=50)                  # many small chunks
    for f, r, h in ((f1, r1, h1), (f2, r2, h2)):
        assert np.array_equal(f, f0) and np.array_equal(r, r0) and np.array_equal(h, h0)
