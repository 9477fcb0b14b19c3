"""Stand-ins for third-party modules the reference imports but this image lacks.

TEST INFRASTRUCTURE ONLY, and only ever used in the build container (the
reference tree does not travel to the GPU box).  ``install()`` pre-seeds
``sys.modules`` so that the *unmodified* files under ``/root/reference``
(nn/conv.py, nn/pool.py, models/cheb_VAE.py, logpdf.py, mesh_operations.py,
config_parser.py, utils.py) import and run; ``oracle/make_golden.py`` then
captures golden vectors from them.

Every stand-in restates the *published* behaviour of the named dependency at
the reference's call sites (requirements.txt pins):

* torch-scatter==2.0.9   ``scatter_add`` / ``scatter(reduce='add'|'sum')``
  (nn/conv.py:363, :551) = zeros(dim_size).scatter_add_(dim, broadcast(index), src)
* torch-geometric==2.0.4 ``utils.remove_self_loops`` (nn/conv.py:544)
  = keep edges with row != col;  ``nn.conv.cheb_conv.ChebConv`` (models/cheb_cls.py:18,75
  -- the crecon classifier's convolution): K bias-free ``Linear`` layers with glorot-uniform
  weights [out, in] (drawn once at construction and again by ChebConv.reset_parameters), a zero
  bias, and ``forward(x, edge_index)`` = sum_k lins[k](T_k) + bias with T_0 = x, T_1 = L^ x,
  T_k = 2 L^ T_{k-1} - T_{k-2}; L^ is built per call as remove_self_loops -> get_laplacian('sym')
  (edges -d^-1/2[row] d^-1/2[col], self loops +1) -> * 2/lambda_max (= 2.0) -> add_self_loops(-1);
  messages are gathered at edge_index[0], summed at edge_index[1] over node_dim = -2
* torch-sparse==0.6.13   ``SparseTensor`` (isinstance check only, nn/conv.py:152)
* open3d (unpinned)      ``io.read_triangle_mesh`` (model.py:36) -> OBJ reader
* psbody-mesh (unpinned) ``Mesh(v=, f=, filename=)`` (model.py:37) and
  ``compute_aabb_tree().nearest(pts, True)`` (mesh_operations.py:208):
  brute-force closest point on triangle with psbody's part codes
  (0 interior, 1/2/3 edges ab/bc/ca, 4/5/6 vertices a/b/c).

Names that are imported by the reference but never reached on the cheb_VAE
path are stubs that raise if called.
"""
import sys
import types

import numpy as np
import torch


# --------------------------------------------------------------------------- torch_scatter
def _broadcast_index(index, src, dim):
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1:
        for _ in range(dim):
            index = index.unsqueeze(0)
    for _ in range(src.dim() - index.dim()):
        index = index.unsqueeze(-1)
    return index.expand_as(src)


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    index = _broadcast_index(index, src, dim)
    if out is None:
        size = list(src.size())
        if dim_size is not None:
            size[dim] = dim_size
        elif index.numel() == 0:
            size[dim] = 0
        else:
            size[dim] = int(index.max()) + 1
        out = torch.zeros(size, dtype=src.dtype, device=src.device)
    return out.scatter_add_(dim, index, src)


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    if reduce not in ("sum", "add"):
        raise NotImplementedError("stand-in covers the reference's reduce='add' call site only")
    return scatter_add(src, index, dim, out, dim_size)


def _unreachable(name):
    def f(*a, **k):
        raise RuntimeError(f"refshim: {name} is not on the cheb_VAE hot path")
    f.__name__ = name
    return f


# --------------------------------------------------------------------------- torch_geometric
def remove_self_loops(edge_index, edge_attr=None):
    mask = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, mask]
    if edge_attr is None:
        return edge_index, None
    return edge_index, edge_attr[mask]


def add_self_loops(edge_index, edge_attr=None, fill_value=1.0, num_nodes=None):
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    loop = torch.arange(0, n, dtype=torch.long, device=edge_index.device).unsqueeze(0).repeat(2, 1)
    if edge_attr is not None:
        fill = edge_attr.new_full((n,) + tuple(edge_attr.shape[1:]), fill_value)
        edge_attr = torch.cat([edge_attr, fill], dim=0)
    return torch.cat([edge_index, loop], dim=1), edge_attr


def get_laplacian(edge_index, edge_weight=None, normalization=None, dtype=None, num_nodes=None):
    if normalization != "sym":
        raise NotImplementedError("stand-in covers ChebConv's default normalization='sym' only")
    edge_index, edge_weight = remove_self_loops(edge_index, edge_weight)
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    row, col = edge_index[0], edge_index[1]
    deg = scatter_add(edge_weight, row, dim=0, dim_size=n)
    deg_inv_sqrt = deg.pow_(-0.5)
    deg_inv_sqrt.masked_fill_(deg_inv_sqrt == float("inf"), 0)
    edge_weight = deg_inv_sqrt[row] * edge_weight * deg_inv_sqrt[col]
    return add_self_loops(edge_index, -edge_weight, fill_value=1.0, num_nodes=n)   # L = I - A_norm


class PygLinear(torch.nn.Module):
    """torch_geometric.nn.dense.linear.Linear(in, out, bias=False, weight_initializer='glorot')."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = torch.nn.Parameter(torch.Tensor(out_channels, in_channels))
        self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = (6.0 / (self.weight.size(-2) + self.weight.size(-1))) ** 0.5      # inits.glorot
        self.weight.data.uniform_(-stdv, stdv)

    def forward(self, x):
        return torch.nn.functional.linear(x, self.weight, self.bias)


class PygChebConv(torch.nn.Module):
    """torch_geometric.nn.conv.ChebConv as published in 2.0.4, 'sym' normalisation, aggr='add',
    flow source_to_target, node_dim=-2 (so [B, N, C] inputs broadcast over B)."""

    def __init__(self, in_channels, out_channels, K, normalization="sym", bias=True):
        super().__init__()
        assert K > 0
        assert normalization in [None, "sym", "rw"], "Invalid normalization"
        self.in_channels, self.out_channels, self.normalization = in_channels, out_channels, normalization
        self.lins = torch.nn.ModuleList([PygLinear(in_channels, out_channels) for _ in range(K)])
        if bias:
            self.bias = torch.nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()
        if self.bias is not None:
            self.bias.data.fill_(0)

    def __norm__(self, edge_index, num_nodes, edge_weight, normalization, lambda_max, dtype=None):
        edge_index, edge_weight = remove_self_loops(edge_index, edge_weight)
        edge_index, edge_weight = get_laplacian(edge_index, edge_weight, normalization, dtype, num_nodes)
        edge_weight = (2.0 * edge_weight) / lambda_max
        edge_weight.masked_fill_(edge_weight == float("inf"), 0)
        return add_self_loops(edge_index, edge_weight, fill_value=-1.0, num_nodes=num_nodes)

    @staticmethod
    def propagate(edge_index, x, norm):
        msg = norm.view(-1, 1) * x.index_select(-2, edge_index[0])
        return scatter_add(msg, edge_index[1], dim=-2, dim_size=x.size(-2))

    def forward(self, x, edge_index, edge_weight=None, batch=None, lambda_max=None):
        if self.normalization != "sym" and lambda_max is None:
            raise ValueError("You need to pass `lambda_max` to `forward() in`"
                             "case the normalization is non-symmetric.")
        if lambda_max is None:
            lambda_max = torch.tensor(2.0, dtype=x.dtype, device=x.device)
        edge_index, norm = self.__norm__(edge_index, x.size(-2), edge_weight, self.normalization,
                                         lambda_max, dtype=x.dtype)
        Tx_0, Tx_1 = x, x
        out = self.lins[0](Tx_0)
        if len(self.lins) > 1:
            Tx_1 = self.propagate(edge_index, x, norm)
            out = out + self.lins[1](Tx_1)
        for lin in self.lins[2:]:
            Tx_2 = self.propagate(edge_index, Tx_1, norm)
            Tx_2 = 2.0 * Tx_2 - Tx_0
            out = out + lin.forward(Tx_2)
            Tx_0, Tx_1 = Tx_1, Tx_2
        if self.bias is not None:
            out += self.bias
        return out


# --------------------------------------------------------------------------- psbody / open3d
def read_obj(path):
    vs, fs = [], []
    with open(path) as fp:
        for line in fp:
            if line.startswith("v "):
                vs.append([float(t) for t in line.split()[1:4]])
            elif line.startswith("f "):
                fs.append([int(t.split("/")[0]) - 1 for t in line.split()[1:4]])
    return np.asarray(vs, dtype=np.float64), np.asarray(fs, dtype=np.int64)


class _O3DMesh:
    def __init__(self, v, f):
        self.vertices = v
        self.triangles = f


def _closest_point_on_triangles(p, a, b, c):
    """Closest point on each triangle (a,b,c)[F] to ONE point p; Ericson RTCD 5.1.5.
    Returns (points[F,3], part[F]) with psbody part codes."""
    ab, ac, ap = b - a, c - a, p - a
    d1 = (ab * ap).sum(-1)
    d2 = (ac * ap).sum(-1)
    bp = p - b
    d3 = (ab * bp).sum(-1)
    d4 = (ac * bp).sum(-1)
    cp = p - c
    d5 = (ab * cp).sum(-1)
    d6 = (ac * cp).sum(-1)
    vc = d1 * d4 - d3 * d2
    vb = d5 * d2 - d1 * d6
    va = d3 * d6 - d5 * d4
    F = a.shape[0]
    pts = np.empty((F, 3))
    part = np.full(F, -1, dtype=np.int64)
    todo = np.ones(F, dtype=bool)

    def take(mask, q, code):
        m = mask & todo
        pts[m] = q[m]
        part[m] = code
        todo[m] = False

    with np.errstate(divide="ignore", invalid="ignore"):
        take((d1 <= 0) & (d2 <= 0), a, 4)
        take((d3 >= 0) & (d4 <= d3), b, 5)
        take((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + (d1 / (d1 - d3))[:, None] * ab, 1)
        take((d6 >= 0) & (d5 <= d6), c, 6)
        take((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + (d2 / (d2 - d6))[:, None] * ac, 3)
        w = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        take((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), b + w[:, None] * (c - b), 2)
        den = 1.0 / (va + vb + vc)
        v = vb * den
        w2 = vc * den
        take(np.ones(F, dtype=bool), a + ab * v[:, None] + ac * w2[:, None], 0)
    return pts, part


class _AabbTree:
    def __init__(self, mesh):
        self.v = np.asarray(mesh.v, dtype=np.float64)
        self.f = np.asarray(mesh.f, dtype=np.int64)

    def nearest(self, pts, nearest_part=False):
        a, b, c = self.v[self.f[:, 0]], self.v[self.f[:, 1]], self.v[self.f[:, 2]]
        pts = np.asarray(pts, dtype=np.float64)
        n = pts.shape[0]
        faces = np.zeros((1, n), dtype=np.uint32)
        parts = np.zeros((1, n), dtype=np.uint32)
        out = np.zeros((n, 3))
        for i in range(n):
            q, part = _closest_point_on_triangles(pts[i], a, b, c)
            d = ((q - pts[i]) ** 2).sum(-1)
            j = int(np.argmin(d))
            faces[0, i], parts[0, i], out[i] = j, part[j], q[j]
        if nearest_part:
            return faces, parts, out
        return faces, out


class Mesh:
    def __init__(self, v=None, f=None, filename=None):
        if filename is not None:
            v, f = read_obj(filename)
        self.v = np.asarray(v, dtype=np.float64)
        self.f = np.asarray(f) if f is not None else None

    def compute_aabb_tree(self):
        return _AabbTree(self)


# --------------------------------------------------------------------------- install
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    if "torch_scatter" in sys.modules and getattr(sys.modules["torch_scatter"], "__refshim__", False):
        return
    _mod("torch_scatter", scatter_add=scatter_add, scatter=scatter,
         gather_csr=_unreachable("gather_csr"), segment_csr=_unreachable("segment_csr"),
         __refshim__=True)
    _mod("torch_sparse", SparseTensor=type("SparseTensor", (), {}))
    tg = _mod("torch_geometric")
    tg.utils = _mod("torch_geometric.utils", remove_self_loops=remove_self_loops,
                    add_self_loops=add_self_loops, degree=_unreachable("degree"),
                    get_laplacian=get_laplacian)
    tg.data = _mod("torch_geometric.data", Dataset=torch.utils.data.Dataset, Data=_unreachable("Data"),
                   DataLoader=_unreachable("DataLoader"))
    if "torchvision" not in sys.modules:      # data.py:9 imports it and never uses it
        tv = _mod("torchvision")
        tv.transforms, tv.utils = _mod("torchvision.transforms"), _mod("torchvision.utils")
    tg.nn = _mod("torch_geometric.nn", dense_diff_pool=_unreachable("dense_diff_pool"),
                 global_sort_pool=_unreachable("global_sort_pool"))
    tg.nn.conv = _mod("torch_geometric.nn.conv")
    tg.nn.conv.cheb_conv = _mod("torch_geometric.nn.conv.cheb_conv", ChebConv=PygChebConv)
    o3d = _mod("open3d")
    o3d.io = _mod("open3d.io", read_triangle_mesh=lambda p: _O3DMesh(*read_obj(p)))
    o3d.geometry = _mod("open3d.geometry")
    ps = _mod("psbody")
    ps.mesh = _mod("psbody.mesh", Mesh=Mesh)
