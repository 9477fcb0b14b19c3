"""`mesh_operations` -- the A / D / U hierarchy of a template mesh (reference mesh_operations.py:13-278,
SURVEY section 8(f) "next" #1), host side, run once per template.

Same entry points and return values as the reference module (`generate_transform_matrices(mesh, factors)
-> (M, A, D, U)` with scipy COO matrices in the entry order model.py:24-32 turns into torch sparse
tensors).  The loops over mesh elements run in C++ (mesh-vae_amd/csrc/host/hierarchy.cpp ->
libmeshvae_host.so, C ABI include/meshvae_host.h, bound below with ctypes):

* adjacency A     : symmetric vertex-vertex incidence accumulated over the three face edges
                    (:13-31), CSC -> COO (column-major entry order; the values count shared faces) -- scipy;
* decimation  D   : Garland-Heckbert quadric edge collapse restricted to vertex pairs (QSlim-style,
                    :87-199).  Face planes: the null vector of [v | 1] by np.linalg.svd, face by face, the
                    reference's own LAPACK call (:58-61) -- on coplanar neighbourhoods (a subdivided template)
                    the collapse order hangs on its last bits.  Everything after that is C++:
                    `mvhh_vertex_quadrics` (face-major accumulation of p p^T), `mvhh_unique_edges` (the queue
                    order), `mvhh_qslim_decimate` (CPython-heapq-compatible heap of mutable edge records, lazy
                    re-evaluation of stale costs, collapse onto the cheaper endpoint, summed quadrics, degenerate
                    faces dropped) until ceil(n * factor) vertices are left; D is the one-hot selection of the
                    surviving vertices in ascending index order (:72-85);
* upsampling  U   : every fine vertex expressed in its closest coarse triangle (:202-250): closest point by
                    `mvhh_closest_points` (bounding-volume hierarchy over the triangles, exact with respect to the
                    all-triangles scan), then barycentric-style weights by np.linalg.lstsq on the triangle's /
                    edge's vertex positions (:229-243, the reference's LAPACK call; a vertex hit gets weight 1).

Measured (8 host cores of the build container; tests/test_mesh_operations.py prints them): 5k template, four
levels 21 s (reference) -> 0.5 s; 20k template, five levels: minutes -> 2 s.  Results are bit-identical to the
reference's generator on the tiny / torus-5k / template-5k / subdivided-20k fixtures (A and D with entry
order, U to 1e-6 of the stand-in closest-point search the fixtures were made with).  There is no numpy
fall-back: without libmeshvae_host.so the calls raise.  The numpy restatement that used to live here is
oracle/hierarchy_oracle.py, the checker of the tests.
"""
import ctypes
import math
import os

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.environ.get("MESHVAE_HOST_LIB") or os.path.join(_HERE, "meshvae_hip", "libmeshvae_host.so")
HOST_ABI_VERSION = 100   # MVHH_ABI_VERSION of include/meshvae_host.h this binding was written against
_lib = None

_D = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_L = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_PL = ctypes.POINTER(ctypes.c_int64)
_I64 = ctypes.c_int64

# name -> (restype, argtypes): every symbol of include/meshvae_host.h (tests/test_host_cpu.py checks the list)
HOST_SIGNATURES = {
    "mvhh_version": (ctypes.c_int32, []),
    "mvhh_unique_edges": (ctypes.c_int32, [_L, _I64, _I64, _L, _PL]),
    "mvhh_vertex_quadrics": (ctypes.c_int32, [_D, _L, _I64, _I64, _D]),
    "mvhh_qslim_decimate": (ctypes.c_int32, [_D, _I64, _L, _I64, _D, _L, _I64, _I64, _L, _PL, _PL, ctypes.c_void_p]),
    "mvhh_closest_points": (ctypes.c_int32, [_D, _I64, _L, _I64, _D, _I64, ctypes.c_int32, _L, _L, _D]),
}


class MeshVaeHostError(RuntimeError):
    pass


class BlasHooks(ctypes.Structure):
    """mvhh_blas_t"""
    _fields_ = [("cblas_dgemv", ctypes.c_void_p), ("cblas_ddot", ctypes.c_void_p), ("ilp64", ctypes.c_int32)]


_blas = False   # False = not looked for yet, None = not found


def numpy_cblas():
    """The cblas_dgemv / cblas_ddot of the BLAS this process's numpy is linked to, as an mvhh_blas_t (or None).

    The reference's pair cost IS two numpy dots (mesh_operations.py:121-122); handing the decimator the very entry
    points numpy dispatches them to makes its costs -- and with them the collapse direction of exactly tied pairs --
    those of the reference run on this machine.  Found through threadpoolctl (the loaded library's path); set
    MESHVAE_HOST_PLAIN_COST=1 to use the library's built-in evaluation order instead."""
    global _blas
    if _blas is False:
        _blas = None
        if os.environ.get("MESHVAE_HOST_PLAIN_COST", "0") not in ("", "0"):
            return _blas
        try:
            from threadpoolctl import threadpool_info
            np.dot(np.ones((1, 2)), np.ones((2, 2)))            # (makes sure the BLAS is loaded)
            # numpy's own BLAS first (scipy ships another OpenBLAS build; np.dot never goes through that one)
            blas_libs = [i for i in threadpool_info() if i.get("user_api") == "blas" and i.get("filepath")]
            blas_libs.sort(key=lambda i: 0 if "numpy" in i["filepath"] else 1)
            for info in blas_libs:
                handle = ctypes.CDLL(info["filepath"])
                for pre, suf, ilp64 in (("scipy_", "64_", 1), ("", "64_", 1), ("scipy_", "", 0), ("", "", 0)):
                    try:
                        gemv = getattr(handle, f"{pre}cblas_dgemv{suf}")
                        dot = getattr(handle, f"{pre}cblas_ddot{suf}")
                    except AttributeError:
                        continue
                    hooks = BlasHooks(ctypes.cast(gemv, ctypes.c_void_p), ctypes.cast(dot, ctypes.c_void_p), ilp64)
                    hooks._keep = handle
                    _blas = hooks
                    return _blas
        except Exception:
            _blas = None
    return _blas


def host_lib():
    """Load libmeshvae_host.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise MeshVaeHostError(f"{HOST_LIB_PATH} is missing: build it with `make -C mesh-vae_amd/csrc host` "
                                   "(or __graft_entry__.build())")
        handle = ctypes.CDLL(HOST_LIB_PATH)
        for name, (res, args) in HOST_SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.mvhh_version() != HOST_ABI_VERSION:
            raise MeshVaeHostError(f"libmeshvae_host.so speaks ABI {handle.mvhh_version()}, this binding {HOST_ABI_VERSION}")
        _lib = handle
    return _lib


def _check(rc, what):
    if rc == -2:
        # the reference's heapq.heappop on an empty queue (mesh_operations.py:148)
        raise IndexError(f"{what}: the edge queue ran dry before the requested vertex count was reached")
    if rc != 0:
        raise MeshVaeHostError(f"{what}: libmeshvae_host error {rc} (index out of range or bad sizes)")


def _f64(a, cols):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if a.ndim != 2 or a.shape[1] != cols:
        raise ValueError(f"expected an [n, {cols}] array, got {a.shape}")
    return a


def _i64(a, cols):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.int64))
    if a.ndim != 2 or a.shape[1] != cols:
        raise ValueError(f"expected an [n, {cols}] index array, got {a.shape}")
    return a


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
    """[E,2] vertex pairs, each undirected edge once with the smaller index first, in the decimator's queue order
    (column-major COO order of the pair matrix, :33-43)."""
    f = _i64(mesh_f, 3)
    out = np.empty((3 * len(f), 2), dtype=np.int64)
    n = ctypes.c_int64(0)
    _check(host_lib().mvhh_unique_edges(f, len(f), len(mesh_v), out, ctypes.byref(n)), "get_vertices_per_edge")
    return out[:n.value].copy()


# --------------------------------------------------------------------------- quadrics
def face_planes(mesh):
    """[F,4] plane (a,b,c,d) of every face with |(a,b,c)| = 1: the null vector of the face's [v | 1] rows by SVD
    and its normalisation, face by face with the SAME numpy entry points the reference uses (:58-61; the stacked
    svd and the axis form of norm agree with them only to the last bits, and those decide collapse ties)."""
    v, f = np.asarray(mesh.v, dtype=np.float64), np.asarray(mesh.f)
    corners = np.concatenate((v[f], np.ones((len(f), 3, 1))), axis=2)       # [F,3,4]: rows (x, y, z, 1)
    planes = np.empty((len(f), 4))
    svd, norm = np.linalg.svd, np.linalg.norm
    for i, m in enumerate(corners):
        p = svd(m)[2][-1].reshape(-1, 1)
        planes[i] = (p / norm(p[0:3])).ravel()
    return planes


def vertex_quadrics(mesh):
    """[N,4,4]: sum over the vertex's faces of p p^T, p = the face's plane (:45-70)."""
    f = _i64(mesh.f, 3)
    n = len(mesh.v)
    q = np.empty((n, 16))
    _check(host_lib().mvhh_vertex_quadrics(face_planes(mesh), f, len(f), n, q), "vertex_quadrics")
    return q.reshape(n, 4, 4)


def _selection_transform(faces, n_original):
    """Renumber the surviving vertices 0..m-1 (ascending old index) and the one-hot [m, n_original] matrix (:72-85)."""
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
    v, f = _f64(mesh.v, 3), _i64(mesh.f, 3)
    n = len(v)
    if n_verts_desired is None:
        n_verts_desired = math.ceil(n * factor)
    q = np.ascontiguousarray(vertex_quadrics(mesh).reshape(n, 16))
    edges = get_vertices_per_edge(v, f)
    kept_faces = np.empty_like(f)
    n_faces, n_coll = ctypes.c_int64(0), ctypes.c_int64(0)
    blas = numpy_cblas()
    _check(host_lib().mvhh_qslim_decimate(v, n, f, len(f), q, edges, len(edges), int(n_verts_desired), kept_faces,
                                          ctypes.byref(n_faces), ctypes.byref(n_coll),
                                          ctypes.byref(blas) if blas is not None else None), "qslim_decimator_transformer")
    return _selection_transform(kept_faces[:n_faces.value], n)


# --------------------------------------------------------------------------- closest point / upsampling
def nearest_on_surface(source, points, exhaustive=False):
    """For every point: (face index, region code, closest point) on the triangle mesh `source`: region 0 interior,
    1/2/3 edge ab/bc/ca, 4/5/6 vertex a/b/c; the first minimum over the faces wins (psbody's AABB-tree `nearest`,
    :208-209, returns this triple).  `exhaustive` tests every triangle instead of walking the hierarchy."""
    sv, sf = _f64(source.v, 3), _i64(source.f, 3)
    pts = _f64(np.asarray(points, dtype=np.float64).reshape(-1, 3), 3)
    n = len(pts)
    face, region, hit = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64), np.zeros((n, 3))
    _check(host_lib().mvhh_closest_points(sv, len(sv), sf, len(sf), pts, n, 1 if exhaustive else 0, face, region, hit),
           "nearest_on_surface")
    return face, region, hit


def setup_deformation_transfer(source, target, use_normals=False):
    """[n_target, n_source] sparse: every target vertex as a combination of the three vertices of its closest
    source triangle (three entries per row, zeros kept explicitly)."""
    sv, sf = np.asarray(source.v, dtype=np.float64), np.asarray(source.f, dtype=np.int64)
    tv = np.asarray(target.v, dtype=np.float64)
    n = tv.shape[0]
    rows, cols, coef = np.zeros(3 * n), np.zeros(3 * n), np.zeros(3 * n)
    face, region, hit = nearest_on_surface(source, tv)
    lstsq = np.linalg.lstsq
    for i in range(n):
        tri = sf[face[i]]
        rows[3 * i:3 * i + 3] = i
        cols[3 * i:3 * i + 3] = tri
        part = int(region[i])
        if part == 0:                                     # inside the triangle: weights of its three vertices
            basis = np.vstack((sv[tri])).T
            coef[3 * i:3 * i + 3] = lstsq(basis, hit[i], rcond=None)[0]
        elif part <= 3:                                   # on an edge: the target itself over the edge's two vertices
            basis = np.vstack((sv[tri[part - 1]], sv[tri[part % 3]])).T
            w = lstsq(basis, tv[i], rcond=None)[0]
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
