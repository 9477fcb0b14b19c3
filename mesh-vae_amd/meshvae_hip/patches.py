"""Vertex-patch plans for the patch ChebConv kernels (csrc/cheb_patch.hip).

The slab kernels of csrc/cheb_lds.hip give a workgroup (mesh, 4 output channels): the whole mesh of ONE channel
quad fits a CU's LDS, but the K*Cin x Cout contraction then has a 4-wide side, which is what keeps it off
v_mfma_f32_16x16x4_f32.  A patch kernel gives a workgroup (mesh, vertex patch) with ALL channels: the recurrence
T_k = 2 L T_{k-1} - T_{k-2} (nn/conv.py:568-572) needs the values of a vertex's neighbours, so a patch carries a
halo of K - 1 rings around the vertices it owns, and ring r is only needed up to order K - 1 - r.  This module cuts
a level's graph into such patches, once, on the host:

  * exclusive sets E_p: a partition of the vertices (recursive bisection per connected component: the 5k hip-bone
    template is TWO bones, 2512 + 2486 vertices, and each bone cuts into two tubes with ~50-vertex rings);
  * the core C_p >= E_p: with a pooling operator U^T fused behind the layer's dX (nn/pool.py:17-20 backward) every
    coarse row is assigned to one patch and the core is closed under the supports of its rows, so that a patch
    can form its rows of U^T dx alone, without atomics; weight gradients sum over E_p only;
  * rings 1..R around the core (breadth-first), local numbering [E_p | C_p \\ E_p | ring 1 | ... | ring R], each
    group by ascending global id, padded to whole 16-vertex tiles;
  * per local vertex `pinfo` = global id | degree << 16 | ring << 24 | exclusive << 28, and the neighbour list in
    padded ELL form with LOCAL ids premultiplied by 5 (the LDS row stride of the kernels in 16-byte units), two per
    word; pad = the zero row behind the last tile.  (The kernels keep a vertex's list in the 16 bytes of padding
    behind its LDS row.)

Nothing here touches values: A / D / U stay the fixtures' (SURVEY 8(c)); a plan changes which workgroup computes a
row and in which order sums are formed (fp32 reassociation, inside the 1e-4 bars of nn/conv.py's contract).
"""
import os

import numpy as np

ROW_STRIDE_16B = 5          # LDS row = 16 floats + 4 floats of padding (bank spread), in 16-byte units
TILE = 16
MAX_DEG = 8
SHORT_DEG = 6               # lists of at most this many neighbours keep their last two slots as pads
LDS_BYTES = 160 * 1024


def lds_bytes(rows16, n_excl=0):
    """LDS a workgroup of the patch kernels needs for a patch of `rows16` (tile-padded) local vertices, `n_excl` of
    them exclusive: rows + lists, -2 / deg, the backward's per-wave dW tiles (8 matrix waves) and db sums."""
    return (rows16 + 1) * ROW_STRIDE_16B * 16 + rows16 * 4 + (8 * 256 + 8 * 16) * 4


def _adjacency(n, rows, cols):
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    keep = rows != cols
    r = np.concatenate([rows[keep], cols[keep]])
    c = np.concatenate([cols[keep], rows[keep]])
    key = np.unique(r * n + c)
    r, c = key // n, key % n
    ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ptr, r + 1, 1)
    ptr = np.cumsum(ptr)
    return ptr, c


def _neighbours(ptr, adj, front):
    if len(front) == 0:
        return np.zeros(0, dtype=np.int64)
    starts, ends = ptr[front], ptr[front + 1]
    total = int((ends - starts).sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    idx = np.repeat(starts - np.concatenate([[0], np.cumsum(ends - starts)[:-1]]), ends - starts) + np.arange(total)
    return np.unique(adj[idx])


def _bfs(ptr, adj, n, src, allowed=None):
    """distances from the vertex set `src` (restricted to allowed == True when given); -1 = not reached"""
    d = np.full(n, -1, dtype=np.int64)
    d[src] = 0
    front = np.atleast_1d(np.asarray(src, dtype=np.int64))
    k = 0
    while len(front):
        k += 1
        nb = _neighbours(ptr, adj, front)
        nb = nb[d[nb] < 0]
        if allowed is not None:
            nb = nb[allowed[nb]]
        d[nb] = k
        front = nb
    return d


def _components(ptr, adj, n, mask):
    lab = np.full(n, -1, dtype=np.int64)
    comps = []
    for v in np.flatnonzero(mask):
        if lab[v] >= 0:
            continue
        d = _bfs(ptr, adj, n, np.array([v]), mask)
        m = np.flatnonzero(d >= 0)
        lab[m] = len(comps)
        comps.append(m)
    return comps


def _order_component(ptr, adj, n, idx):
    """an ordering of the connected vertex set idx along its longest direction"""
    if len(idx) < 3:
        return idx
    try:                                  # Fiedler vector (scipy is a dependency of this package already)
        if len(idx) < 64:
            raise ValueError
        import scipy.sparse as sp
        import scipy.sparse.linalg as sla
        loc = np.full(n, -1, dtype=np.int64)
        loc[idx] = np.arange(len(idx))
        rr = np.repeat(idx, ptr[idx + 1] - ptr[idx])
        cc = np.concatenate([adj[ptr[v]:ptr[v + 1]] for v in idx])
        keep = loc[cc] >= 0
        a = sp.coo_matrix((np.ones(int(keep.sum())), (loc[rr[keep]], loc[cc[keep]])), shape=(len(idx),) * 2).tocsr()
        lap = (sp.diags(np.asarray(a.sum(1)).ravel()) - a).astype(np.float64)
        vals, vecs = sla.eigsh(lap, k=2, sigma=-1e-4, which="LM", v0=np.ones(len(idx)))
        f = vecs[:, int(np.argsort(vals)[1])]
        if f[int(np.argmax(np.abs(f)))] < 0:
            f = -f
        return idx[np.argsort(f, kind="stable")]
    except Exception:                     # two-source breadth-first ordering (no eigen-solver needed)
        mask = np.zeros(n, dtype=bool)
        mask[idx] = True
        d0 = _bfs(ptr, adj, n, idx[:1], mask)
        a = idx[int(np.argmax(d0[idx]))]
        da = _bfs(ptr, adj, n, np.array([a]), mask)
        b = idx[int(np.argmax(da[idx]))]
        db = _bfs(ptr, adj, n, np.array([b]), mask)
        key = (da[idx] - db[idx]) * (4 * n) + da[idx]
        return idx[np.argsort(key, kind="stable")]


def _split(ptr, adj, n, idx, m):
    """idx (any vertex set) into m parts of equal size, cut across the long direction of its components"""
    if m <= 1:
        return [np.sort(idx)]
    mask = np.zeros(n, dtype=bool)
    mask[idx] = True
    comps = sorted(_components(ptr, adj, n, mask), key=lambda c: (-len(c), int(c[0])))
    order = np.concatenate([_order_component(ptr, adj, n, c) for c in comps])
    m1 = m // 2
    cut = int(round(len(idx) * m1 / m))
    return _split(ptr, adj, n, order[:cut], m1) + _split(ptr, adj, n, order[cut:], m - m1)


def partition(n, ptr, adj, n_parts):
    """-> list of n_parts sorted vertex arrays covering 0..n-1 exactly once"""
    comps = sorted(_components(ptr, adj, n, np.ones(n, dtype=bool)), key=lambda c: (-len(c), int(c[0])))
    big = [c for c in comps if len(c) * 4 * n_parts >= n][:n_parts] or comps[:1]
    small = [c for c in comps if not any(c is b for b in big)]
    sizes = np.array([len(c) for c in big], dtype=np.float64)
    alloc = np.maximum(1, np.floor(sizes / sizes.sum() * n_parts).astype(int))
    while alloc.sum() > n_parts:
        alloc[int(np.argmax(alloc))] -= 1
    while alloc.sum() < n_parts:
        alloc[int(np.argmax(sizes / alloc))] += 1
    parts = []
    for c, m in zip(big, alloc):
        parts += _split(ptr, adj, n, c, int(m))
    for c in small:                       # crumbs (isolated vertices, tiny pieces): whole, to the lightest part
        i = int(np.argmin([len(p) for p in parts]))
        parts[i] = np.sort(np.concatenate([parts[i], c]))
    return parts


class PatchPlan:
    """Host arrays of one plan (see the module docstring); `device(dev)` uploads them and fills mvh_patch_plan_t."""

    def __init__(self, n, n_rings, parts, ptr, adj, pool_t=None):
        self.n, self.n_rings, self.n_patches = int(n), int(n_rings), len(parts)
        deg = (ptr[1:] - ptr[:-1]).astype(np.int64)
        owner = np.full(n, -1, dtype=np.int64)
        for p, e in enumerate(parts):
            owner[e] = p
        assert (owner >= 0).all()
        rows_of = [[] for _ in parts]
        self.n_pool_rows = 0
        if pool_t is not None:
            prp, pcl, pvl = (np.asarray(a) for a in pool_t)
            self.n_pool_rows = len(prp) - 1
            for r in range(self.n_pool_rows):
                sup = pcl[prp[r]:prp[r + 1]]
                if len(sup) == 0:
                    rows_of[0].append(r)       # an empty row is a row of zeros: anyone can write it
                    continue
                votes = np.bincount(owner[sup], minlength=len(parts))
                rows_of[int(np.argmax(votes))].append(r)
        poff, cnt, pinfo, ell = [0], [], [], []
        prow_off, prow_gid, prow_ptr, pcol, pval = [0], [], [], [], []
        self.local_of = []
        for p, e in enumerate(parts):
            core = set(e.tolist())
            for r in rows_of[p]:
                core.update(pcl[prp[r]:prp[r + 1]].tolist())
            core = np.array(sorted(core), dtype=np.int64)
            ring = _bfs_rings(ptr, adj, n, core, n_rings)
            excl = np.zeros(n, dtype=bool)
            excl[e] = True
            groups = [e, core[~excl[core]]] + [np.flatnonzero(ring == r) for r in range(1, n_rings + 1)]
            # inside a group: vertices of <= 6 neighbours first (their lists end in two pads: the kernels skip the last
            # two gathers of a tile whose 16 lists all do -- 81 % of the 5k template's vertices), then by global id
            groups = [g[np.lexsort((g, deg[g] > SHORT_DEG))] for g in groups]
            local = np.concatenate(groups)
            tot = len(local)
            tot16 = (tot + TILE - 1) // TILE * TILE
            loc = np.full(n, -1, dtype=np.int64)
            loc[local] = np.arange(tot)
            c = [len(e), len(core)]
            for r in range(1, n_rings + 1):
                c.append(c[-1] + len(groups[1 + r]))
            cnt.append(c)
            info = np.zeros(tot16, dtype=np.uint32)
            rg = np.concatenate([np.zeros(len(core), dtype=np.int64)] +
                                [np.full(len(groups[1 + r]), r, dtype=np.int64) for r in range(1, n_rings + 1)])
            info[:tot] = (local | (deg[local] << 16) | (rg << 24) | (excl[local].astype(np.int64) << 28)).astype(np.uint32)
            info[tot:] = 15 << 24
            slots = np.full((tot16, MAX_DEG), tot16, dtype=np.int64)
            for li in range(c[-2] if n_rings >= 1 else 0):       # the outermost ring is never gathered FOR
                v = local[li]
                nb = loc[adj[ptr[v]:ptr[v + 1]]]
                assert (nb >= 0).all(), "patch plan: a neighbour of an inner-ring vertex is outside the patch"
                slots[li, :len(nb)] = nb
            slots = _conflict_aware_slots(slots, tot16, c[-2] if n_rings >= 1 else 0)
            if os.environ.get("MESHVAE_PLAN_TIMING_ONLY") == "own_rows":
                # TIMING ONLY, results invalid (tools/microbench_conv.py): every list entry names the vertex's OWN row, the
                # pads stay -- no gather of a tile can conflict; what the kernels would cost with perfect lists
                real = slots != tot16
                slots = np.where(real, np.arange(tot16)[:, None], tot16)
            slots *= ROW_STRIDE_16B
            ell.append((slots[:, 0::2] | (slots[:, 1::2] << 16)).astype(np.uint32))
            pinfo.append(info)
            poff.append(poff[-1] + tot16)
            self.local_of.append(loc)
            # rows of U^T this patch forms: CSR in the operator's own order, LOCAL column ids
            pp = [len(pcol)]
            for r in rows_of[p]:
                cc = loc[pcl[prp[r]:prp[r + 1]]]
                assert (cc >= 0).all() and (cc < c[1]).all()
                pcol += cc.tolist()
                pval += pvl[prp[r]:prp[r + 1]].tolist()
                pp.append(len(pcol))
            prow_gid += rows_of[p]
            prow_ptr += pp
            prow_off.append(prow_off[-1] + len(rows_of[p]))
        self.parts = parts
        self.poff = np.asarray(poff, dtype=np.int32)
        self.cnt = np.asarray(cnt, dtype=np.int32)                  # [P][n_rings + 2]
        self.pinfo = np.concatenate(pinfo)
        self.ell = np.concatenate(ell)                              # [poff[P]][4]
        self.prow_off = np.asarray(prow_off, dtype=np.int32)
        self.prow_gid = np.asarray(prow_gid, dtype=np.int32)
        self.prow_ptr = np.asarray(prow_ptr, dtype=np.int32)        # patch p: entries prow_off[p] + p .. (+ rows_p + 1)
        self.pcol = np.asarray(pcol, dtype=np.int32)
        self.pval = np.asarray(pval, dtype=np.float32)
        self.max_rows = int(max(self.poff[1:] - self.poff[:-1]))
        self.max_pool_nnz = max([int(self.prow_ptr[self.prow_off[p + 1] + p] - self.prow_ptr[self.prow_off[p] + p])
                                 for p in range(self.n_patches)] or [0]) if len(self.prow_ptr) else 0
        self._dev = {}

    def lds_bytes(self):
        return lds_bytes(self.max_rows, int(self.cnt[:, 0].max()))

    def work_ratio(self, K):
        """vertex-orders the patches compute / vertex-orders of the mesh (>= 1: the halo's redundancy)"""
        tot = 0
        for c in self.cnt:
            tot += sum(int(c[1 + min(self.n_rings, K - 1 - k)]) for k in range(1, K))
        return tot / max(1, self.n * (K - 1))

    def attach_unpool(self, rowptr, col, val, n_coarse):
        """U itself (forward CSR: rows = this level's vertices, columns = the coarse level's), for the forward kernel's
        un-pooling loads: urec[slot] = three (coarse row, weight bits) pairs in U's entry order.  Only when every row has at
        most three entries (barycentric up-sampling); otherwise the plan carries none and the caller un-pools first."""
        rowptr, col, val = np.asarray(rowptr, dtype=np.int64), np.asarray(col, dtype=np.int64), np.asarray(val, dtype=np.float32)
        self.urec, self.u_rows = None, 0
        if len(rowptr) != self.n + 1 or (rowptr[1:] - rowptr[:-1]).max() > 3 or n_coarse <= 0:
            return False
        gid = (self.pinfo & 0xffff).astype(np.int64)
        live = ((self.pinfo >> 24) & 15) != 15
        rec = np.zeros((len(self.pinfo), 6), dtype=np.uint32)
        for j in range(3):
            e = rowptr[gid] + j
            ok = live & (e < rowptr[gid + 1])
            ee = np.where(ok, e, 0)
            rec[:, 2 * j] = np.where(ok, col[ee], 0).astype(np.uint32)
            rec[:, 2 * j + 1] = np.where(ok, val[ee], np.float32(0)).astype(np.float32).view(np.uint32)
        self.urec, self.u_rows = rec, int(n_coarse)
        self._dev.clear()
        return True

    def device(self, dev, pool_rowptr=None):
        """-> (PatchPlanStruct, tensors kept alive) on torch device `dev`; pool_rowptr: the device rowptr tensor of the
        pooling operator the plan was built from (the kernels' identity check)"""
        key = (str(dev), None if pool_rowptr is None else pool_rowptr.data_ptr())
        if key not in self._dev:
            import torch
            from . import PatchPlanStruct
            t = {k: torch.from_numpy(np.ascontiguousarray(getattr(self, k)).view(
                     np.int32 if getattr(self, k).dtype == np.uint32 else getattr(self, k).dtype)).to(dev)
                 for k in ("poff", "cnt", "pinfo", "ell", "prow_off", "prow_gid", "prow_ptr", "pcol", "pval")}
            for k in ("prow_gid", "prow_ptr", "pcol", "pval"):     # (never empty pointers: one dummy element)
                if t[k].numel() == 0:
                    t[k] = torch.zeros(1, dtype=t[k].dtype, device=dev)
            urec = getattr(self, "urec", None)
            if urec is not None:
                t["urec"] = torch.from_numpy(np.ascontiguousarray(urec).view(np.int32)).to(dev)
            s = PatchPlanStruct(self.n_patches, self.n_rings, self.n, self.max_rows, int(self.cnt[:, 1].max()),
                                int(self.cnt[:, 0].max()), self.n_pool_rows, int(self.cnt[:, 1].min()),
                                t["poff"].data_ptr(), t["cnt"].data_ptr(), t["pinfo"].data_ptr(), t["ell"].data_ptr(),
                                t["prow_off"].data_ptr(), t["prow_gid"].data_ptr(), t["prow_ptr"].data_ptr(),
                                t["pcol"].data_ptr(), t["pval"].data_ptr(),
                                None if pool_rowptr is None else pool_rowptr.data_ptr(), self.max_pool_nnz,
                                getattr(self, "u_rows", 0) if urec is not None else 0,
                                t["urec"].data_ptr() if urec is not None else None)
            t["pool_rowptr"] = pool_rowptr
            self._dev[key] = (s, t)
        return self._dev[key]


_H_LANES = (0, 1, 2, 3, 12, 13, 14, 15)          # tile vertices whose quad-0 lanes share a ds_read_b128 lane group


def _conflict_aware_slots(slots, pad, n_gather, rounds=3):
    """Permute every vertex's neighbour list (numpy [rows16, 8] local ids, `pad` = the zero row) against LDS bank
    conflicts of the kernels' gathers.  A wave reads slot j of the 16 vertices of a tile with one ds_read_b128: lane
    (vertex i, quad q) fetches the 16 bytes at (5 n + q) * 16, and the instruction is served in four 16-lane groups
    (MI355X_MICROARCH.md, LDS): quads 0 / 1 of the tile's vertices {0-3, 12-15} / {4-11}, the same with the two vertex
    sets swapped, and both again for quads 2 / 3.  A group costs one LDS cycle per distinct 16-byte word in its most
    loaded 4-bank column ((5 n + q) mod 16); equal words broadcast.  The sums are unweighted (scaled variables), so the
    order of a vertex's neighbours is free: coordinate descent over the tile's vertices, each step an exact 8 x 8
    assignment of the vertex's ids to the slots against what the 15 others read.  Model cycles per gather on the 5k
    template: 1.9 -> ~1.2."""
    from scipy.optimize import linear_sum_assignment
    out = slots.copy()
    S = slots.shape[1]
    for t0 in range(0, n_gather, TILE):
        vs = [v for v in range(t0, min(t0 + TILE, slots.shape[0]))]
        rows = [out[v].tolist() for v in vs]
        in_h = [(v - t0) in _H_LANES for v in vs]
        # per slot j and condition (0: H reads quad 0 / M quad 1; 1: swapped): column -> {word: readers}
        use = [[[dict() for _ in range(16)] for _ in range(2)] for _ in range(S)]

        def cols(n, h):             # the vertex's word column under the two conditions
            a = (5 * n) % 16
            return (a, (a + 1) % 16) if h else ((a + 1) % 16, a)

        def book(i, row, sign):
            for j, n in enumerate(row):
                for cnd, col in enumerate(cols(n, in_h[i])):
                    d = use[j][cnd][col]
                    k = d.get(n, 0) + sign
                    if k:
                        d[n] = k
                    else:
                        del d[n]
        for i, r in enumerate(rows):
            book(i, r, +1)
        for _ in range(rounds):
            changed = False
            for i, r in enumerate(rows):
                if all(n == pad for n in r):
                    continue
                book(i, r, -1)
                real = [n for n in r if n != pad]
                # a short list keeps its neighbours in the first SHORT_DEG slots (the last two stay pads)
                n_slots = SHORT_DEG if len(real) <= SHORT_DEG else S
                cost = np.zeros((len(real), n_slots))
                for a_, n in enumerate(real):
                    c0, c1 = cols(n, in_h[i])
                    for j in range(n_slots):
                        d0, d1 = use[j][0][c0], use[j][1][c1]
                        cost[a_, j] = (0 if (not d0 or n in d0) else len(d0)) + (0 if (not d1 or n in d1) else len(d1))
                ri, ci = linear_sum_assignment(cost)
                new = [pad] * S
                for a_, j in zip(ri, ci):
                    new[j] = real[a_]
                changed |= new != r
                rows[i] = new
                book(i, new, +1)
            if not changed:
                break
        out[vs] = np.asarray(rows, dtype=slots.dtype)
    return out


def gather_conflict_model(plan):
    """mean LDS cycles of the eight (or six: see SHORT_DEG) gather instructions of one tile under the model above;
    a conflict-free full tile costs 8 instructions x 4 lane groups = 32"""
    tot, tiles = 0.0, 0
    for p in range(plan.n_patches):
        o, rows16 = int(plan.poff[p]), int(plan.poff[p + 1] - plan.poff[p])
        e = plan.ell[o:o + rows16].astype(np.int64)
        nb = np.stack([e & 0xffff, e >> 16], -1).reshape(rows16, 8) // ROW_STRIDE_16B
        n_g = int(plan.cnt[p][-2])
        for t0 in range(0, n_g, TILE):
            short = bool((nb[t0:t0 + TILE, SHORT_DEG:] == rows16).all())
            for j in range(SHORT_DEG if short else 8):
                for cnd in range(2):
                    binned = {}
                    for i in range(TILE):
                        n = int(nb[t0 + i, j])
                        a = (5 * n) % 16
                        h = i in _H_LANES
                        col = (a if (h ^ (cnd == 1)) else (a + 1) % 16)
                        binned.setdefault(col, set()).add(n)
                    tot += 2 * max(len(v) for v in binned.values())      # (quads 2 / 3 repeat the pattern of 0 / 1)
            tiles += 1
    return tot / max(tiles, 1)


def _bfs_rings(ptr, adj, n, core, n_rings):
    d = np.full(n, -1, dtype=np.int64)
    d[core] = 0
    front = core
    for r in range(1, n_rings + 1):
        nb = _neighbours(ptr, adj, front)
        nb = nb[d[nb] < 0]
        d[nb] = r
        front = nb
    return d


_plan_cache = {}


def build_plan(n, rows, cols, n_rings, pool_t=None, min_patches=None, max_patches=10, lds_cap=LDS_BYTES):
    """The plan with the fewest patches (>= min_patches, default ceil(n / 1300)) whose largest patch fits the LDS
    budget, or None: vertices of more than 8 neighbours, or no cut of <= max_patches patches fits (expander-like
    graphs: the halo of 5 rings is the whole graph) -- such a level stays on the slab kernels.
    rows / cols: the edge list of the level (any order, both directions or one); pool_t: (rowptr, col, val) of a
    pooling operator's TRANSPOSE (rows = coarse vertices, columns = this level's vertices) or None."""
    rows = np.asarray(rows)
    cols = np.asarray(cols)
    key = (int(n), int(n_rings), hash(rows.tobytes()), hash(cols.tobytes()),
           None if pool_t is None else tuple(hash(np.asarray(a).tobytes()) for a in pool_t), min_patches, max_patches, lds_cap)
    if key in _plan_cache:
        return _plan_cache[key]
    plan = None
    ptr, adj = _adjacency(n, rows, cols)
    if n >= 2 * TILE and int((ptr[1:] - ptr[:-1]).max()) <= MAX_DEG and n < 65536:
        p0 = min_patches if min_patches is not None else max(1, -(-n // 1280))    # (exclusive sets of <= 1 296 vertices: the kernels' group slots)
        for n_parts in range(p0, max_patches + 1):
            parts = partition(n, ptr, adj, n_parts)
            cand = PatchPlan(n, n_rings, parts, ptr, adj, pool_t)
            if cand.lds_bytes() <= lds_cap and cand.max_rows * ROW_STRIDE_16B < 65536:
                plan = cand
                break
    _plan_cache[key] = plan
    return plan


# ---------------------------------------------------------------------------------------------------------------------
# numpy model of what the kernels compute from a plan (tests: the plan's index structure against a dense ChebConv)
def emulate_forward(plan, x, W, bias=None, relu=False):
    """x [n, Cin] (one mesh), W [K, Cin, Cout] -> out [n, Cout], computed patch by patch exactly as
    k_patch_fwd does: scaled variables u = D^-1/2 T, ring-limited orders, outputs of the exclusive vertices."""
    K = W.shape[0]
    out = np.zeros((plan.n, W.shape[2]), dtype=np.float64)
    for p in range(plan.n_patches):
        o, c = int(plan.poff[p]), plan.cnt[p]
        rows16 = int(plan.poff[p + 1]) - o
        info = plan.pinfo[o:o + rows16]
        gid, deg = (info & 0xffff).astype(np.int64), ((info >> 16) & 0xff).astype(np.float64)
        s = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1)), 1.0)
        coef = np.where(deg > 0, -2.0 / np.maximum(deg, 1), 0.0)
        e = plan.ell[o:o + rows16].astype(np.int64)
        nb = np.stack([e & 0xffff, e >> 16], -1).reshape(rows16, 8) // ROW_STRIDE_16B     # local ids, pad = rows16
        tot = int(c[1 + plan.n_rings])
        u = np.zeros((rows16 + 1, x.shape[1]))
        n0 = int(c[1 + min(plan.n_rings, K - 1)])
        u[:n0] = x[gid[:n0]] * s[:n0, None]
        prev = np.zeros_like(u)
        acc = u[:rows16] @ W[0]
        for k in range(1, K):
            nk = int(c[1 + min(plan.n_rings, K - 1 - k)])
            new = np.zeros_like(u)
            g = u[nb[:nk]].sum(1)
            new[:nk] = (0.5 if k == 1 else 1.0) * coef[:nk, None] * g - prev[:nk]
            prev, u = u, new
            acc += u[:rows16] @ W[k]
        ne = int(c[0])
        res = acc[:ne] / s[:ne, None]
        if bias is not None:
            res = res + bias
        if relu:
            res = np.maximum(res, 0)
        out[gid[:ne]] = res
        assert tot <= rows16
    return out
