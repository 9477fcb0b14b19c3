"""CPU restatement of the reference's hierarchy generator in numpy.  TEST INFRASTRUCTURE ONLY.

The checker for `mesh-vae_amd/mesh_operations.py` (whose element loops run in C++, libmeshvae_host.so):
only tests/ may import this module; the product never does.  It follows reference
mesh_operations.py:13-31 (adjacency), :45-70 (quadrics), :72-85 (_get_sparse_transform), :87-199 (decimator),
:202-250 (deformation transfer), :253-278 (generate_transform_matrices), written from the algorithm:

* adjacency A     : symmetric vertex-vertex incidence accumulated over the three face edges, CSC -> COO;
* decimation  D   : Garland-Heckbert quadric edge collapse restricted to vertex pairs: per-vertex quadrics from
                    the faces' normalised plane equations (plane = null vector of [v | 1] by SVD), a min-heap of
                    edges keyed by min(q_sum(v_r), q_sum(v_c)), lazy re-evaluation of stale costs, collapse onto
                    the cheaper endpoint without moving it, both endpoints inherit the summed quadric, degenerate
                    faces dropped, until ceil(n * factor) vertices are left;
* upsampling  U   : every fine vertex expressed in its closest coarse triangle: closest point on the coarse
                    surface, then barycentric-style weights by least squares.

Parity pin: tests/test_mesh_operations.py holds it to the A / D / U hierarchies captured from the reference's own
generator (tests/golden/topology_{tiny,5k,20k}.npz, hier_torus5k.npz; generators oracle/make_golden*.py):
adjacency and decimation bit-identical, entry order included.  The closest-point search is pinned only against
the brute-force stand-in of oracle/refshim (psbody's AABB tree is absent from the image).

The heap holds small mutable edge records that every live endpoint indexes (the reference rescans its whole
queue twice per collapse); the pair cost is evaluated through numpy's dot exactly as the reference writes it.
"""
import heapq
import math

import numpy as np
import scipy.sparse as sp


# --------------------------------------------------------------------------- mesh holder / OBJ
def read_obj(path):
    """Vertices [N,3] float64 and triangles [F,3] int64 of a Wavefront OBJ (`v` and `f` records)."""
    verts, faces = [], []
    with open(path) as fp:
        for line in fp:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                verts.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif tok[0] == "f":
                faces.append(tuple(int(t.split("/")[0]) - 1 for t in tok[1:4]))
    return np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int64)


class Mesh:
    """The two attributes of psbody.mesh.Mesh the hierarchy code touches: `.v` [N,3], `.f` [F,3]."""

    def __init__(self, v=None, f=None, filename=None):
        if filename is not None:
            v, f = read_obj(filename)
        self.v = np.asarray(v, dtype=np.float64)
        self.f = None if f is None else np.asarray(f)


# --------------------------------------------------------------------------- adjacency
def get_vert_connectivity(mesh_v, mesh_f):
    """[N,N] sparse CSC, entry (i,j) = number of (directed) face edges joining i and j, both ways."""
    n = len(mesh_v)
    f = np.asarray(mesh_f)
    acc = sp.csc_matrix((n, n))
    for k in range(3):
        a, b = f[:, k], f[:, (k + 1) % 3]
        one_way = sp.csc_matrix((np.ones(len(a)), np.vstack((a.reshape(1, -1), b.reshape(1, -1)))), shape=(n, n))
        acc = acc + one_way + one_way.T
    return acc


def get_vertices_per_edge(mesh_v, mesh_f):
    """[E,2] vertex pairs, each undirected edge once with the smaller index first."""
    coo = sp.coo_matrix(get_vert_connectivity(mesh_v, mesh_f))
    pairs = np.hstack((coo.row.reshape(-1, 1), coo.col.reshape(-1, 1)))
    return pairs[pairs[:, 0] < pairs[:, 1]]


# --------------------------------------------------------------------------- quadrics
def vertex_quadrics(mesh):
    """[N,4,4]: sum over the vertex's faces of p p^T, p = the face's plane (a,b,c,d) with |(a,b,c)| = 1."""
    v, f = np.asarray(mesh.v, dtype=np.float64), np.asarray(mesh.f)
    corners = np.concatenate((v[f], np.ones((len(f), 3, 1))), axis=2)       # [F,3,4]: rows (x, y, z, 1)
    # Null vector of each 3x4 system and its normalisation, face by face with the SAME numpy entry points the
    # reference uses (one np.linalg.svd and one vector np.linalg.norm, a BLAS dot, per face): the stacked
    # svd and the axis form of norm agree with them only to the last bits, and on (nearly) coplanar
    # neighbourhoods -- a subdivided template -- the collapse order is decided by exactly those bits.
    planes = np.empty((len(f), 4))
    for i, m in enumerate(corners):
        p = np.linalg.svd(m)[2][-1].reshape(-1, 1)
        planes[i] = (p / np.linalg.norm(p[0:3])).ravel()
    outer = planes[:, :, None] * planes[:, None, :]
    q = np.zeros((len(v), 4, 4))
    for face, pp in zip(f, outer):          # face-major accumulation: the order fixes the last bits of the sums,
        for vert in face:                   # and the collapse order is decided by comparing them
            q[vert] += pp
    return q


class _Edge:
    """Heap record: ordered like the tuple (cost, (r, c)); r / c are rewritten in place on collapses."""
    __slots__ = ("cost", "r", "c")

    def __init__(self, cost, r, c):
        self.cost, self.r, self.c = cost, r, c

    def __lt__(self, other):
        if self.cost != other.cost:
            return self.cost < other.cost
        if self.r != other.r:
            return self.r < other.r
        return self.c < other.c


def _pair_costs(q, r, c, v):
    """Quadric error of keeping r's position / keeping c's position for the merged pair, and the summed quadric."""
    qs = q[r] + q[c]
    pr = np.append(v[r], 1.0).reshape(-1, 1)
    pc = np.append(v[c], 1.0).reshape(-1, 1)
    keep_r = float(pr.T.dot(qs).dot(pr)[0, 0])      # error if c is destroyed
    keep_c = float(pc.T.dot(qs).dot(pc)[0, 0])      # error if r is destroyed
    return keep_r, keep_c, qs


def _selection_transform(faces, n_original):
    """Renumber the surviving vertices 0..m-1 (ascending old index) and the one-hot [m, n_original] matrix."""
    left = np.unique(faces.ravel())
    remap = np.arange(0, np.max(faces.ravel()) + 1)
    remap[left] = np.arange(len(left))
    new_faces = remap[faces.ravel()].reshape(-1, 3)
    ij = np.vstack((np.arange(len(left)), left))
    return new_faces, sp.csc_matrix((np.ones(len(left)), ij), shape=(len(left), n_original))


def qslim_decimator_transformer(mesh, factor=None, n_verts_desired=None):
    """-> (new_faces [F',3], D sparse [n', n]) keeping ceil(n * factor) (or n_verts_desired) vertices."""
    if factor is None and n_verts_desired is None:
        raise Exception('Need either factor or n_verts_desired.')
    v = np.asarray(mesh.v, dtype=np.float64)
    n = len(v)
    if n_verts_desired is None:
        n_verts_desired = math.ceil(n * factor)
    q = vertex_quadrics(mesh)

    # undirected edges in the reference's queue order: COO entries of the symmetrised pair matrix with r <= c
    pairs = get_vertices_per_edge(v, mesh.f)
    adj = sp.csc_matrix((pairs[:, 0] * 0 + 1, (pairs[:, 0], pairs[:, 1])), shape=(n, n))
    adj = (adj + adj.T).tocoo()
    heap, touching = [], [[] for _ in range(n)]
    for r, c in zip(adj.row, adj.col):
        if r > c:
            continue
        keep_r, keep_c, _ = _pair_costs(q, r, c, v)
        e = _Edge(keep_c if keep_c < keep_r else keep_r, int(r), int(c))
        heapq.heappush(heap, e)
        touching[r].append(e)
        touching[c].append(e)

    faces = np.asarray(mesh.f).copy()
    live = np.ones(len(faces), dtype=bool)
    uses = np.bincount(faces.ravel(), minlength=n)
    n_left = int(np.count_nonzero(uses))
    while n_left > n_verts_desired:
        e = heapq.heappop(heap)
        r, c = e.r, e.c
        if r == c:
            continue
        keep_r, keep_c, qs = _pair_costs(q, r, c, v)
        now = keep_c if keep_c < keep_r else keep_r
        if now > e.cost:                       # stale: the endpoints' quadrics grew since it was queued
            fresh = _Edge(now, r, c)
            heapq.heappush(heap, fresh)
            touching[r].append(fresh)
            touching[c].append(fresh)
            continue
        gone, kept = (c, r) if keep_r < keep_c else (r, c)
        # faces: rename the vertex, drop what became degenerate
        hit = live & np.any(faces == gone, axis=1)
        rows = np.nonzero(hit)[0]
        sub = faces[rows]
        k = int(np.count_nonzero(sub == gone))
        sub[sub == gone] = kept
        faces[rows] = sub
        uses[kept] += k
        if uses[gone] > 0:
            n_left -= 1
        uses[gone] = 0
        dead = (sub[:, 0] == sub[:, 1]) | (sub[:, 1] == sub[:, 2]) | (sub[:, 2] == sub[:, 0])
        if dead.any():
            live[rows[dead]] = False
            for vert in sub[dead].ravel():
                uses[vert] -= 1
                if uses[vert] == 0:
                    n_left -= 1
        # queue: every record that mentions the vanished vertex now mentions the kept one
        for rec in touching[gone]:
            if rec.r == gone:
                rec.r = kept
            if rec.c == gone:
                rec.c = kept
        touching[kept].extend(touching[gone])
        touching[gone] = []
        q[r] = qs
        q[c] = qs
    return _selection_transform(faces[live], n)


# --------------------------------------------------------------------------- closest point / upsampling
def _closest_on_triangles(p, a, ab, ac, b, c):
    """Closest point of every triangle (a, a+ab, a+ac) to the point p, with the region code of psbody's
    AABB query: 0 interior, 1/2/3 edge ab/bc/ca, 4/5/6 vertex a/b/c (Ericson, Real-Time Collision Detection 5.1.5)."""
    ap, bp, cp = p - a, p - b, p - c
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    m = a.shape[0]
    out = np.empty((m, 3))
    code = np.full(m, -1, dtype=np.int64)
    open_ = np.ones(m, dtype=bool)

    def settle(mask, pts, region):
        sel = mask & open_
        out[sel] = pts[sel]
        code[sel] = region
        open_[sel] = False

    with np.errstate(divide="ignore", invalid="ignore"):
        settle((d1 <= 0) & (d2 <= 0), a, 4)
        settle((d3 >= 0) & (d4 <= d3), b, 5)
        settle((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + (d1 / (d1 - d3))[:, None] * ab, 1)
        settle((d6 >= 0) & (d5 <= d6), c, 6)
        settle((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + (d2 / (d2 - d6))[:, None] * ac, 3)
        w = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        settle((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), b + w[:, None] * (c - b), 2)
        den = 1.0 / (va + vb + vc)
        settle(np.ones(m, dtype=bool), a + ab * (vb * den)[:, None] + ac * (vc * den)[:, None], 0)
    return out, code


def _nearest_exhaustive(points, a, ab, ac, b, c):
    n = len(points)
    face, region, hit = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64), np.zeros((n, 3))
    for i in range(n):
        pts, code = _closest_on_triangles(points[i], a, ab, ac, b, c)
        j = int(np.argmin(((pts - points[i]) ** 2).sum(-1)))
        face[i], region[i], hit[i] = j, code[j], pts[j]
    return face, region, hit


def nearest_on_surface(source, points, chunk_pairs=1 << 21):
    """For every point: (face index, region code, closest point) on the triangle mesh `source` (first minimum wins).

    Same result as testing every triangle for every point (the arithmetic per (point, triangle) pair is
    elementwise and unchanged), but only the triangles that can hold the minimum are tested: the distance
    d0 to the nearest mesh vertex bounds the distance to the surface, and a triangle within d0 of the point
    has its centroid within d0 + r_max (r_max = largest centroid-to-corner distance).  Candidates are
    visited in ascending face order, so ties resolve to the lowest face index as in the exhaustive scan.
    20k-vertex template: 108 s -> ~2 s."""
    from scipy.spatial import cKDTree
    sv, sf = np.asarray(source.v, dtype=np.float64), np.asarray(source.f, dtype=np.int64)
    a, b, c = sv[sf[:, 0]], sv[sf[:, 1]], sv[sf[:, 2]]
    ab, ac = b - a, c - a
    points = np.asarray(points, dtype=np.float64)
    n = len(points)
    area2 = (np.cross(ab, ac) ** 2).sum(-1)
    if n == 0 or len(sf) < 64 or not np.all(np.isfinite(area2)) or np.any(area2 == 0):
        return _nearest_exhaustive(points, a, ab, ac, b, c)     # tiny or degenerate input: no pruning
    cen = (a + b + c) / 3.0
    r_max = float(np.sqrt(max(((a - cen) ** 2).sum(-1).max(), ((b - cen) ** 2).sum(-1).max(),
                              ((c - cen) ** 2).sum(-1).max())))
    d0, _ = cKDTree(sv[np.unique(sf)]).query(points)
    radius = (d0 + r_max) * (1.0 + 1e-9) + 1e-12 * r_max
    cand = cKDTree(cen).query_ball_point(points, radius, return_sorted=True)
    counts = np.fromiter((len(t) for t in cand), dtype=np.int64, count=n)
    face, region, hit = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64), np.zeros((n, 3))
    lo = 0
    while lo < n:                                         # chunks of whole points, bounded pair count
        hi, tot = lo, 0
        while hi < n and (hi == lo or tot + counts[hi] <= chunk_pairs):
            tot += counts[hi]
            hi += 1
        pf = np.concatenate([np.asarray(cand[i], dtype=np.int64) for i in range(lo, hi)])
        pp = np.repeat(np.arange(lo, hi), counts[lo:hi])
        pts, code = _closest_on_triangles(points[pp], a[pf], ab[pf], ac[pf], b[pf], c[pf])
        d = ((pts - points[pp]) ** 2).sum(-1)
        if np.any(np.isnan(d)):
            return _nearest_exhaustive(points, a, ab, ac, b, c)
        starts = np.concatenate(([0], np.cumsum(counts[lo:hi])[:-1]))
        seg_min = np.minimum.reduceat(d, starts)
        at_min = np.flatnonzero(d == np.repeat(seg_min, counts[lo:hi]))
        owner = pp[at_min]
        first = at_min[np.concatenate(([True], owner[1:] != owner[:-1]))]
        face[lo:hi], region[lo:hi], hit[lo:hi] = pf[first], code[first], pts[first]
        lo = hi
    return face, region, hit


def setup_deformation_transfer(source, target, use_normals=False):
    """[n_target, n_source] sparse: every target vertex as a combination of the three vertices of its closest
    source triangle (three entries per row, zeros kept explicitly)."""
    sv, sf = np.asarray(source.v, dtype=np.float64), np.asarray(source.f, dtype=np.int64)
    tv = np.asarray(target.v, dtype=np.float64)
    n = tv.shape[0]
    rows, cols, coef = np.zeros(3 * n), np.zeros(3 * n), np.zeros(3 * n)
    face, region, hit = nearest_on_surface(source, tv)
    for i in range(n):
        tri = sf[face[i]]
        rows[3 * i:3 * i + 3] = i
        cols[3 * i:3 * i + 3] = tri
        part = int(region[i])
        if part == 0:                                     # inside the triangle: weights of its three vertices
            basis = np.vstack((sv[tri])).T
            coef[3 * i:3 * i + 3] = np.linalg.lstsq(basis, hit[i], rcond=None)[0]
        elif part <= 3:                                   # on an edge: the target itself over the edge's two vertices
            basis = np.vstack((sv[tri[part - 1]], sv[tri[part % 3]])).T
            w = np.linalg.lstsq(basis, tv[i], rcond=None)[0]
            coef[3 * i + part - 1] = w[0]
            coef[3 * i + part % 3] = w[1]
        else:                                             # at a vertex
            coef[3 * i + part - 4] = 1.0
    return sp.csc_matrix((coef, (rows, cols)), shape=(n, sv.shape[0]))


# --------------------------------------------------------------------------- the hierarchy
def generate_transform_matrices(mesh, factors):
    """(M, A, D, U): meshes decimated by 1/factors[i] in turn, their adjacencies, and the down / up transforms."""
    M, A, D, U = [mesh], [get_vert_connectivity(mesh.v, mesh.f).tocoo()], [], []
    for factor in factors:
        new_f, down = qslim_decimator_transformer(M[-1], factor=1.0 / factor)
        D.append(down.tocoo())
        coarse = Mesh(v=down.dot(M[-1].v), f=new_f)
        M.append(coarse)
        A.append(get_vert_connectivity(coarse.v, coarse.f).tocoo())
        U.append(setup_deformation_transfer(M[-1], M[-2]).tocoo())
    return M, A, D, U
