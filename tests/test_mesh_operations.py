"""CPU: the build's hierarchy generator (mesh-vae_amd/mesh_operations.py over the C++ library
libmeshvae_host.so) against the A / D / U hierarchies captured from the reference's own
generator (tests/golden/topology_*.npz): adjacency and decimation exactly (entry order included), upsampling
weights to 1e-6 (its closest-point search is the stand-in's in both cases, psbody being unpinned -- see
DESIGN.md section 2), and piece by piece against the numpy restatement oracle/hierarchy_oracle.py."""
import os
import time

import numpy as np
import pytest

import mesh_operations as mo
from conftest import load_golden
from oracle import hierarchy_oracle as ho
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
    the case where the heap's tie-breaking and the last bits of the quadrics decide the result: the decimator is given numpy's own CBLAS entry points (mesh_operations.numpy_cblas) so that the
    costs of tied pairs carry the reference's last bits."""
    t = mo.Mesh(filename=REF_TEMPLATE)
    v, f = subdivide(t.v, t.f)
    t0 = time.time()
    M, A, D, U = mo.generate_transform_matrices(mo.Mesh(v=v, f=f), [4, 4, 4, 4, 4])
    print(f"20k hierarchy in {time.time() - t0:.1f} s")
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


def _closest_point_cases(mesh, g, n_each=40):
    v = mesh.v
    pick = g.choice(len(v), 3 * n_each, replace=False)
    span = np.ptp(v, axis=0).max()
    mid = 0.5 * (v[mesh.f[:n_each, 0]] + v[mesh.f[:n_each, 1]])
    return np.concatenate([v[pick[:n_each]],                                                     # exactly on vertices (ties)
                           mid,                                                                    # exactly on edges (two faces tie)
                           v[pick[n_each:2 * n_each]] + g.standard_normal((n_each, 3)) * 0.01 * span,  # near the surface
                           v[pick[2 * n_each:]] + g.standard_normal((n_each, 3)) * 2.0 * span])    # far away


def test_bvh_closest_point_equals_exhaustive_scan_and_numpy_oracle():
    """mvhh_closest_points: the hierarchy walk returns exactly what its own all-triangles scan and the numpy oracle's
    scan return -- faces, region codes and hit points bit for bit, for points on, near and far from the surface
    (ties resolve to the lowest face)."""
    mesh = mo.Mesh(*torus_mesh(51, 98))
    pts = _closest_point_cases(mesh, np.random.default_rng(0))
    sf = np.asarray(mesh.f, dtype=np.int64)
    a, b, c = mesh.v[sf[:, 0]], mesh.v[sf[:, 1]], mesh.v[sf[:, 2]]
    f0, r0, h0 = ho._nearest_exhaustive(pts, a, b - a, c - a, b, c)
    f1, r1, h1 = mo.nearest_on_surface(mesh, pts)
    f2, r2, h2 = mo.nearest_on_surface(mesh, pts, exhaustive=True)
    assert len(set(r0.tolist())) >= 4            # interior, edge and vertex regions all occur
    for f, r, h in ((f1, r1, h1), (f2, r2, h2)):
        assert np.array_equal(f, f0) and np.array_equal(r, r0) and np.array_equal(h, h0)


def test_closest_point_degenerate_triangles_follow_the_full_scan():
    """A zero-area triangle makes distances NaN for some points; the library then takes the all-triangles scan with
    numpy's argmin rule (the first NaN wins) -- same answers as the oracle's exhaustive search."""
    v, f = torus_mesh(9, 12)
    f = np.concatenate([f[:5], [[3, 3, 3]], f[5:]])            # a point triangle in the middle of the list
    mesh = mo.Mesh(v, f)
    pts = _closest_point_cases(mo.Mesh(v, f[:5]), np.random.default_rng(1), n_each=5)
    sf = np.asarray(f, dtype=np.int64)
    a, b, c = v[sf[:, 0]], v[sf[:, 1]], v[sf[:, 2]]
    with np.errstate(all="ignore"):
        f0, r0, h0 = ho._nearest_exhaustive(pts, a, b - a, c - a, b, c)
    f1, r1, h1 = mo.nearest_on_surface(mesh, pts)
    assert np.array_equal(f1, f0) and np.array_equal(r1, r0) and np.array_equal(h1, h0, equal_nan=True)


def test_decimator_pieces_against_the_numpy_oracle():
    """Edge queue order, quadrics, surviving faces (order and rotation included) and D of one decimation, C++ against
    the numpy restatement, on the torus; with the caller's CBLAS and with the library's built-in cost arithmetic."""
    v, f = torus_mesh(21, 30)
    mesh, omesh = mo.Mesh(v, f), ho.Mesh(v, f)
    assert np.array_equal(mo.get_vertices_per_edge(v, f), ho.get_vertices_per_edge(v, f))
    assert np.array_equal(mo.vertex_quadrics(mesh), ho.vertex_quadrics(omesh))
    nf0, d0 = ho.qslim_decimator_transformer(omesh, factor=0.25)
    nf1, d1 = mo.qslim_decimator_transformer(mesh, factor=0.25)
    assert mo.numpy_cblas() is not None, "numpy's CBLAS entry points were not found (threadpoolctl)"
    assert np.array_equal(nf1, nf0) and np.array_equal(d1.tocoo().col, d0.tocoo().col)
    nf2, d2 = mo.qslim_decimator_transformer(mesh, n_verts_desired=d0.shape[0])
    assert np.array_equal(nf2, nf0)
    old, mo._blas = mo._blas, None                              # the built-in evaluation order (no BLAS hooks)
    try:
        nf3, d3 = mo.qslim_decimator_transformer(mesh, factor=0.25)
    finally:
        mo._blas = old
    assert np.array_equal(d3.tocoo().col, d0.tocoo().col) and np.array_equal(nf3, nf0)


def test_decimator_runs_dry_like_the_reference():
    """A vertex count the mesh cannot reach (every face gone and still above the target) ends in the reference's
    IndexError (heappop on an empty queue, mesh_operations.py:148), not in a hang or a wrong answer."""
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float64)
    f = np.array([[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]])
    with pytest.raises(IndexError):
        mo.qslim_decimator_transformer(mo.Mesh(v, f), n_verts_desired=-1)
    with pytest.raises(IndexError):
        ho.qslim_decimator_transformer(ho.Mesh(v, f), n_verts_desired=-1)


def test_integration_md_host_stub_runs_as_written():
    """The ctypes stub INTEGRATION.md shows for libmeshvae_host.so, executed as written (only the library path is made
    absolute): its surviving faces are the product binding's and the oracle's."""
    import re
    from conftest import ROOT
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = [c for c in re.findall(r"```python\n(.*?)```", text, re.S) if "libmeshvae_host.so" in c][0]
    code = code.replace('"/path/to/repo/mesh-vae_amd/meshvae_hip/libmeshvae_host.so"', repr(mo.HOST_LIB_PATH))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    v, f = torus_mesh(21, 30)
    mesh = mo.Mesh(v, f)
    old, mo._blas = mo._blas, None                    # (the stub passes no BLAS hooks)
    try:
        nf, d = mo.qslim_decimator_transformer(mesh, n_verts_desired=160)
    finally:
        mo._blas = old
    faces = ns["qslim_faces"](np.ascontiguousarray(v), np.ascontiguousarray(f, dtype=np.int64), mo.face_planes(mesh), 160)
    nf2, d2 = mo._selection_transform(faces, len(v))
    assert np.array_equal(nf2, nf) and np.array_equal(d2.tocoo().col, d.tocoo().col)
    nf0, _ = ho.qslim_decimator_transformer(ho.Mesh(v, f), n_verts_desired=160)
    assert np.array_equal(nf, nf0)


def _jittered_torus(nu, nv, seed, amp):
    """A torus grid with every vertex moved by a seeded random offset (generic geometry: no exact cost ties), faces
    unchanged."""
    v, f = torus_mesh(nu, nv)
    g = np.random.default_rng(seed)
    span = np.ptp(v, axis=0).max()
    return v + g.standard_normal(v.shape) * amp * span / max(nu, nv), f


@pytest.mark.parametrize("nu,nv,seed,amp,factor", [(9, 14, 1, 0.2, 0.5), (12, 17, 2, 0.05, 0.25), (16, 23, 3, 0.3, 0.125),
                                                   (7, 31, 4, 0.1, 0.34), (20, 20, 5, 0.0, 0.25)])
def test_random_meshes_cpp_hierarchy_equals_numpy_oracle(nu, nv, seed, amp, factor):
    """Whole levels on seeded random geometry, C++ against the numpy restatement: the surviving faces (order and rotation),
    D, the closest-point triples of the fine vertices on the coarse surface, and U entry for entry (amp = 0: the regular
    torus, where whole rings of edges have EQUAL costs up to rounding -- the heap's tie-breaking decides)."""
    v, f = _jittered_torus(nu, nv, seed, amp)
    fine, ofine = mo.Mesh(v, f), ho.Mesh(v, f)
    nf0, d0 = ho.qslim_decimator_transformer(ofine, factor=factor)
    nf1, d1 = mo.qslim_decimator_transformer(fine, factor=factor)
    assert np.array_equal(nf1, nf0)
    assert np.array_equal(d1.tocoo().col, d0.tocoo().col) and d1.shape == d0.shape
    coarse = mo.Mesh(v=d1.dot(v), f=nf1)
    f1, r1, h1 = mo.nearest_on_surface(coarse, v)
    f0, r0, h0 = ho.nearest_on_surface(ho.Mesh(v=coarse.v, f=coarse.f), v)
    assert np.array_equal(f1, f0) and np.array_equal(r1, r0) and np.array_equal(h1, h0)
    u1 = mo.setup_deformation_transfer(coarse, fine).tocoo()
    u0 = ho.setup_deformation_transfer(ho.Mesh(v=coarse.v, f=coarse.f), ofine).tocoo()
    assert np.array_equal(u1.row, u0.row) and np.array_equal(u1.col, u0.col) and np.array_equal(u1.data, u0.data)
    # the defining property of U D on the kept vertices: a kept fine vertex is reproduced from itself
    kept = d1.tocoo().col
    back = u1.tocsr()[kept].dot(coarse.v)
    np.testing.assert_allclose(back, v[kept], rtol=0, atol=1e-9 * np.ptp(v))
