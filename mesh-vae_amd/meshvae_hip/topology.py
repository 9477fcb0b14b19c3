"""Fixed-topology preprocessing: the reference's COO edge lists -> device CSR.

The reference hands the model *uncoalesced* COO tensors (model.py:24-32) and walks them
edge by edge every call (nn/conv.py:199-200, :363).  Here each operator is converted once
into CSR over output rows (stable, so a row keeps the reference's edge order and therefore
its fp32 accumulation order) plus the CSR of its transpose for the backward pass, int32
indices, uploaded once and cached for the life of the tensors.
"""
import ctypes

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from . import CSR_ELL_OVERFLOW, CSR_NORMALIZED_LAPLACIAN, CSR_SELECTION, CSR_SYMMETRIC, CsrStruct


# ds_read_b128 serves a wave in four fixed 16-lane groups, one LDS cycle per group when no two lanes of a group
# hit different 16-byte words of the same 4-bank column (MI355X_MICROARCH.md, LDS): lanes l of a wave <-> vertices
# v = l (mod 64) in the thread-owns-vertex layout of the LDS-resident ChebConv kernels, word column = neighbour id mod 16.
_B128_GROUPS = [np.array(g) for g in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                                      [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31])]
_B128_GROUPS = _B128_GROUPS + [g + 32 for g in _B128_GROUPS]


# ds_read_b64 (the float2 planes of csrc/cheb_big.hip): two 32-lane halves, 8-byte word column = neighbour id mod 32
_B64_GROUPS = [np.arange(0, 32), np.arange(32, 64)]
_SLOT_CACHE = {}


def conflict_aware_slots(slots, pad, rounds=4, groups=None, ncol=16):
    """Permute every vertex's neighbour list (numpy [n_rows, S], `pad` = filler id) so that, for each lane group of
    the gather (default: the four 16-lane groups of ds_read_b128) and each list slot j, the ids read together fall into
    distinct word columns (id mod ncol) as far as possible; equal ids broadcast for free.  The order of a vertex's neighbours is irrelevant to the unweighted sums
    of the LDS kernels (L is applied in scaled variables), so this is free at run time: on the 5k template the model
    count of LDS cycles per gather instruction drops from 1.99 to 1.15 (1.0 = conflict-free), levels 1-3 alike.
    Method: coordinate descent, one vertex at a time, each step an exact 8 x 8 assignment (Hungarian) of its ids to
    the slots against what the other 15 vertices of its group currently read."""
    n_rows, S = slots.shape
    groups = _B128_GROUPS if groups is None else groups
    key = (slots.shape, int(pad), rounds, ncol, len(groups), hash(slots.tobytes()))
    if key in _SLOT_CACHE:                       # (a process builds the same template many times: tests, micro-batches)
        return _SLOT_CACHE[key].copy()
    out = slots.copy()
    for w0 in range(0, n_rows, 64):
        for g in groups:
            vs = g + w0
            vs = vs[vs < n_rows]
            if vs.size < 2:
                continue
            rows = [out[v].tolist() for v in vs]
            usage = [[dict() for _ in range(ncol)] for _ in range(S)]    # slot -> column -> {id: readers}

            def book(row, sign):
                for j, u in enumerate(row):
                    d = usage[j][u % ncol]
                    n = d.get(u, 0) + sign
                    if n:
                        d[u] = n
                    else:
                        del d[u]
            for r in rows:
                book(r, +1)
            for _ in range(rounds):
                changed = False
                for i, r in enumerate(rows):
                    book(r, -1)
                    cost = np.zeros((S, S))
                    for a, u in enumerate(r):
                        for j in range(S):
                            d = usage[j][u % ncol]
                            cost[a, j] = 0 if (not d or u in d) else len(d)
                    ri, ci = linear_sum_assignment(cost)
                    new = [pad] * S
                    for a, j in zip(ri, ci):
                        new[j] = r[a]
                    changed |= new != r
                    rows[i] = new
                    book(new, +1)
                if not changed:
                    break
            out[vs] = np.asarray(rows, dtype=slots.dtype)
    _SLOT_CACHE[key] = out.copy()
    return out


class Csr:
    """One CSR operator resident on the device + its mvh_csr_t descriptor."""

    def __init__(self, out_idx, in_idx, val, n_rows, n_cols, device):
        out_idx = out_idx.detach().to("cpu", torch.int64)
        in_idx = in_idx.detach().to("cpu", torch.int64)
        val = val.detach().to("cpu", torch.float32)
        if out_idx.numel():
            if int(out_idx.min()) < 0 or int(out_idx.max()) >= n_rows:
                raise ValueError(f"sparse operator row index out of range [0, {n_rows})")
            if int(in_idx.min()) < 0 or int(in_idx.max()) >= n_cols:
                raise ValueError(f"sparse operator column index out of range [0, {n_cols})")
        order = torch.sort(out_idx, stable=True).indices
        counts = torch.bincount(out_idx, minlength=n_rows)
        rowptr = torch.zeros(n_rows + 1, dtype=torch.int64)
        rowptr[1:] = torch.cumsum(counts, 0)
        self.n_rows, self.n_cols, self.nnz = int(n_rows), int(n_cols), int(out_idx.numel())
        self.max_row_nnz = int(counts.max()) if counts.numel() else 0
        col_sorted, val_sorted = in_idx[order], val[order].contiguous()
        self.rowptr = rowptr.to(torch.int32).to(device)
        self.col = col_sorted.to(torch.int32).to(device)
        self.val = val_sorted.to(device)
        # ---- compact form + structure flags for the LDS-resident kernels (host analysis, once)
        self.flags = 0
        self.rowinfo = None
        if self.n_rows == self.n_cols and self.nnz > 0:
            deg = counts.to(torch.float32)
            dis = deg.pow(-0.5)
            dis[dis == float("inf")] = 0
            rows_sorted = out_idx[order]
            if torch.equal(val_sorted, -dis[col_sorted] * torch.ones_like(val_sorted) * dis[rows_sorted]):
                self.flags |= CSR_NORMALIZED_LAPLACIAN
            fwd_key = rows_sorted * self.n_cols + col_sorted
            bwd_key = col_sorted * self.n_cols + rows_sorted
            fo, bo = torch.sort(fwd_key).indices, torch.sort(bwd_key).indices
            if torch.equal(fwd_key[fo], bwd_key[bo]) and torch.equal(val_sorted[fo], val_sorted[bo]):
                self.flags |= CSR_SYMMETRIC
        self.ell, self.ell_pairs = None, 0
        if self.n_cols < 65535 and 0 < self.max_row_nnz < 256 and self.nnz < (1 << 24):
            info = ((rowptr[:-1] << 8) | counts).numpy().astype("uint32")      # bit pattern kept in int32
            self.rowinfo = torch.from_numpy(info.view("int32")).to(device)
            self.ell_pairs = (self.max_row_nnz + 1) // 2
            pos = torch.arange(self.nnz) - rowptr[:-1][out_idx[order]]          # position inside the row
            keep = torch.ones(self.nnz, dtype=torch.bool)
            if 8 < self.max_row_nnz <= 12:
                # a few long rows (2-6 % of a decimated level) would double the padded width for every
                # vertex: the list keeps the first 8 columns, the kernels fetch the rest from `col`
                self.flags |= CSR_ELL_OVERFLOW
                self.ell_pairs = 4
                keep = pos < 8
            pw = 4 if self.ell_pairs <= 4 else 8 * ((self.ell_pairs + 7) // 8)   # words per vertex (16-byte groups)
            slots = torch.full((self.n_rows, 2 * pw), self.n_cols, dtype=torch.int64)
            slots[out_idx[order][keep], pos[keep]] = col_sorted[keep]
            if self.n_rows == self.n_cols and self.n_rows + 1 <= 5120 and (self.flags & CSR_NORMALIZED_LAPLACIAN):
                # (only the LDS-resident kernels read the list, and only where edges carry no values)
                slots = torch.from_numpy(conflict_aware_slots(slots.numpy(), self.n_cols))
            elif self.n_rows == self.n_cols and self.n_rows <= 20480 and pw == 4 and \
                    (self.flags & CSR_NORMALIZED_LAPLACIAN) and not (self.flags & CSR_ELL_OVERFLOW):
                # csrc/cheb_big.hip: float2 planes gathered with ds_read_b64 (32-lane halves, id mod 32)
                slots = torch.from_numpy(conflict_aware_slots(slots.numpy(), self.n_cols, rounds=2,
                                                              groups=_B64_GROUPS, ncol=32))
            packed = (slots[:, 0::2] | (slots[:, 1::2] << 16)).contiguous()     # [n_rows, pw] vertex-major
            self.ell = torch.from_numpy(packed.numpy().astype("uint32").view("int32")).to(device)
        self.struct = CsrStruct(self.n_rows, self.n_cols, self.nnz, self.rowptr.data_ptr(),
                                self.col.data_ptr(), self.val.data_ptr(),
                                self.rowinfo.data_ptr() if self.rowinfo is not None else None,
                                self.ell.data_ptr() if self.ell is not None else None,
                                self.ell_pairs, self.max_row_nnz, self.flags, 0, None, None)
        # one-hot selection (downsampling D): one unit entry per row, no repeated column
        self.sel_inv = None
        if self.nnz == self.n_rows and self.nnz > 0 and bool((counts == 1).all()) and bool((val_sorted == 1.0).all()) \
                and int(torch.unique(col_sorted).numel()) == self.nnz:
            inv = torch.full((self.n_cols,), -1, dtype=torch.int32)
            inv[col_sorted] = out_idx[order].to(torch.int32)
            self.sel_inv = inv.to(device)
            self.flags |= CSR_SELECTION
            self.struct.flags = self.flags
            self.struct.sel_inv = self.sel_inv.data_ptr()
        self.n_active = int(max(int(out_idx.max()), int(in_idx.max())) + 1) if self.nnz else 0
        self.struct.n_active = self.n_active
        self.sub = None

    def attach_sub(self, sub):
        """`sub` = this operator restricted to its leading n_active x n_active block."""
        self.sub = sub
        self.struct.sub = ctypes.addressof(sub.struct)

    @property
    def ref(self):
        return ctypes.byref(self.struct)


class Operator:
    """A sparse operator y = P x together with P^T (for the backward pass)."""

    def __init__(self, out_idx, in_idx, val, n_out, n_in, device):
        self.fwd = Csr(out_idx, in_idx, val, n_out, n_in, device)
        self.bwd = Csr(in_idx, out_idx, val, n_in, n_out, device)
        self.n_out, self.n_in = int(n_out), int(n_in)


_cache = {}


def _key(*tensors, extra=()):
    return tuple((t.data_ptr(), t._version, tuple(t.shape), str(t.device)) for t in tensors) + tuple(extra)


def laplacian(edge_index, norm, num_nodes):
    """CSR of the propagate of ChebConv_batch (flow source_to_target, nn/conv.py:172):
    gather at edge_index[0], sum at edge_index[1]; square [num_nodes, num_nodes] with
    empty rows where the edge list does not reach (cheb_VAE.py:288)."""
    if edge_index.dim() != 2 or edge_index.size(0) != 2 or edge_index.dtype != torch.long:
        raise ValueError("`edge_index` must be a torch.LongTensor of shape [2, num_messages]")
    k = _key(edge_index, norm, extra=(int(num_nodes),))
    hit = _cache.get(k)
    if hit is None:
        op = Operator(edge_index[1], edge_index[0], norm, num_nodes, num_nodes, norm.device)
        na = op.fwd.n_active
        if 0 < na and 4 * na <= num_nodes:
            # mostly-isolated graph (the final layer's coarsest edge list on the finest vertices):
            # hand the kernels the small connected problem as well
            sub = Operator(edge_index[1], edge_index[0], norm, na, na, norm.device)
            op.fwd.attach_sub(sub.fwd)
            op.bwd.attach_sub(sub.bwd)
        # a level the vertex-patch kernels take (csrc/cheb_patch.hip): 16 -> 16 layers of up to K = 6 run as (mesh, patch)
        # workgroups; plain plan (no fused pooling) for the module-level calls -- the step engine attaches its own
        got = patch_plan(op, 5)
        if got is not None:
            import ctypes
            op.fwd.struct.patch = op.bwd.struct.patch = ctypes.addressof(got[0])
        hit = _cache[k] = (op, edge_index, norm)   # keep the tensors alive: data_ptr is the key
    return hit[0]


def patch_plan(lap_op, n_rings, up_op=None, down_op=None):
    """Vertex-patch plan (meshvae_hip/patches.py, csrc/cheb_patch.hip) of a level's Laplacian, uploaded:
    -> (PatchPlanStruct, keep-alive) or None when the level is not one the patch kernels take (2 049 .. 5 119 vertices,
    normalised symmetric Laplacian, at most 8 neighbours, a cut whose largest patch fits the LDS).  up_op: the level's
    un-pooling Operator (coarse -> this level); its transpose's rows are then formed inside the backward kernel.
    down_op: the level's one-hot downsampling Operator (this level -> coarse) instead: its rows are the pooled rows the
    first layer's kernel (k_patch_enc0) contracts and stores."""
    import ctypes
    from . import patches
    csr = lap_op.fwd
    need = CSR_NORMALIZED_LAPLACIAN | CSR_SYMMETRIC
    if (csr.flags & need) != need or not (2048 < csr.n_rows + 1 <= 5120) or csr.max_row_nnz > patches.MAX_DEG or n_rings < 0:
        return None
    assert up_op is None or down_op is None
    cache = lap_op.__dict__.setdefault("_patch_plans", {})
    key = (int(n_rings), None if up_op is None else id(up_op), None if down_op is None else id(down_op))
    if key not in cache:
        rowptr = csr.rowptr.cpu().numpy().astype(np.int64)
        rows = np.repeat(np.arange(csr.n_rows), rowptr[1:] - rowptr[:-1])
        cols = csr.col.cpu().numpy().astype(np.int64)
        pool_t, pool_rowptr = None, None
        t = None
        if up_op is not None and up_op.bwd.n_cols == csr.n_rows:
            t = up_op.bwd                                     # U^T: rows = coarse vertices, columns = this level
        elif down_op is not None and down_op.fwd.n_cols == csr.n_rows:
            t = down_op.fwd                                   # D: rows = coarse vertices, columns = this level
        if t is not None:
            pool_t = (t.rowptr.cpu().numpy().astype(np.int64), t.col.cpu().numpy().astype(np.int64), t.val.cpu().numpy())
            pool_rowptr = t.rowptr
        plan = patches.build_plan(csr.n_rows, rows, cols, int(n_rings), pool_t)
        # a cut into small patches is all halo: below ~900 exclusive vertices per patch (K = 10 on a 5k level: 8 patches of
        # 625 whose 9 rings make 1632 rows each, 1.5 x the row-orders and two rounds of workgroups at B = 64) the slab
        # kernels are faster -- no plan then
        if plan is not None and csr.n_rows / plan.n_patches < 900:
            plan = None
        if plan is not None and up_op is not None and up_op.fwd.n_rows == csr.n_rows and getattr(plan, "urec", None) is None:
            u = up_op.fwd                                     # U: rows = this level, columns = the coarse level
            plan.attach_unpool(u.rowptr.cpu().numpy(), u.col.cpu().numpy(), u.val.cpu().numpy(), u.n_cols)
        cache[key] = None if plan is None else plan.device(csr.rowptr.device, pool_rowptr) + (plan, up_op if up_op is not None else down_op)
    return cache[key]


def pool_operator(pool_mat):
    """CSR of SurfacePool's propagate (flow target_to_source, nn/pool.py:15,19):
    out row = indices[0], gathered input row = indices[1], size = pool_mat.size()."""
    idx, val = pool_mat._indices(), pool_mat._values()
    k = _key(idx, val, extra=tuple(pool_mat.size()))
    hit = _cache.get(k)
    if hit is None:
        n_out, n_in = pool_mat.size()
        op = Operator(idx[0], idx[1], val, n_out, n_in, val.device)
        hit = _cache[k] = (op, idx, val)
    return hit[0]


def clear_cache():
    _cache.clear()
