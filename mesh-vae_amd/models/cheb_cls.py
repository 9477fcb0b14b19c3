"""`models.cheb_cls` -- MI355X-native contrastive-reconstruction classifier.

Mirror of the reference module of the same name (models/cheb_cls.py:22-114): `Pool`, `cheb_GCN`
with the reference's constructor, attributes, forward and `reset_parameters`, and -- because the
reference takes its convolution from torch-geometric (cheb_cls.py:18) -- a `ChebConv` with that
package's 2.0.4 parameter layout (`lins.{k}.weight` [C_out, C_in] per Chebyshev order, `bias`),
initialisation (glorot-uniform, zero bias) and RNG consumption, so that `torch.manual_seed(s)` gives
bit-identical initial weights and reference checkpoints load unchanged.

The arithmetic is the same as the VAE encoder's: torch-geometric's scaled 'sym' Laplacian with
lambda_max = 2 is -D^-1/2 A D^-1/2 plus self-loop entries +1 and -1 that cancel (the older in-tree copy
of the operator: nn/conv.py:464-484), i.e. exactly the operator `ChebConv_batch.norm` builds, so every
layer runs libmeshvae_hip's LDS-resident ChebConv kernels (fused ReLU, sign bytes for the backward)
and the one-hot gather of `mvh_pool_fwd`.  The +x - x rounding noise of the published operator
(<= 1 ulp of x per order) is not reproduced; parity is within the 1e-4 fp32 tolerance of the hot
path (tests/test_gpu_parity.py).  The unused `graph_norm` and `CNN` classes of the reference file are
dead code there and are not built.
"""
import math

import torch
import torch.nn.functional as F

from meshvae_hip import functional as F_hip
from meshvae_hip import topology
from nn.conv import ChebConv_batch


def Pool(x, trans, dim=1):
    """trans @ x per mesh for a sparse COO `trans` [N_out, N_in] (cheb_cls.py:22-27)."""
    if dim != 1:
        raise NotImplementedError("Pool: only the reference's call shape [B, N, C], dim=1 is built")
    return F_hip.surface_pool(x, topology.pool_operator(trans))


class Linear(torch.nn.Module):
    """Bias-free linear layer with glorot-uniform weights: torch-geometric's
    Linear(in, out, bias=False, weight_initializer='glorot') as ChebConv builds it."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = torch.nn.Parameter(torch.Tensor(out_channels, in_channels))
        self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        bound = math.sqrt(6.0 / (self.weight.size(-2) + self.weight.size(-1)))
        self.weight.data.uniform_(-bound, bound)

    def forward(self, x):
        lead = x.shape[:-1]
        return F_hip.linear(x.reshape(-1, self.in_channels), self.weight, None).reshape(*lead, self.out_channels)


# input widths the LDS-resident kernels are instantiated for; anything narrower is zero-padded up
_LDS_WIDTHS = (3, 8, 16, 32)


class ChebConv(torch.nn.Module):
    """torch-geometric-compatible Chebyshev convolution on [B, N, C] (or [N, C]) inputs.

    out = sum_k lins[k](T_k) + bias, T_0 = x, T_1 = L^ x, T_k = 2 L^ T_{k-1} - T_{k-2} with
    L^ = -D^-1/2 A D^-1/2 of `edge_index` (self loops dropped; 'sym' normalisation, lambda_max = 2).
    """

    def __init__(self, in_channels, out_channels, K, normalization='sym', bias=True, **kwargs):
        super().__init__()
        assert K > 0
        assert normalization in [None, 'sym', 'rw'], 'Invalid normalization'
        self.in_channels, self.out_channels, self.normalization = in_channels, out_channels, normalization
        self.lins = torch.nn.ModuleList([Linear(in_channels, out_channels) for _ in range(K)])
        if bias:
            self.bias = torch.nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()
        if self.bias is not None:
            self.bias.data.fill_(0)

    def _laplacian(self, edge_index, n):
        key = (edge_index.data_ptr(), edge_index.shape[1], n, str(edge_index.device))
        hit = getattr(self, '_lap_cache', None)
        if hit is None or hit[0] != key:
            ei, norm = ChebConv_batch.norm(edge_index, n, dtype=torch.float32)
            hit = (key, edge_index, topology.laplacian(ei, norm, n))      # keeps edge_index alive with its key
            self._lap_cache = hit
        return hit[2]

    def forward(self, x, edge_index, edge_weight=None, batch=None, lambda_max=None, relu=False):
        if self.normalization != 'sym':
            if lambda_max is None:
                raise ValueError('You need to pass `lambda_max` to `forward() in`'
                                 'case the normalization is non-symmetric.')
            raise NotImplementedError("ChebConv: only normalization='sym' (the reference's call) is built")
        if edge_weight is not None or batch is not None or (lambda_max is not None and float(lambda_max) != 2.0):
            raise NotImplementedError("ChebConv: only forward(x, edge_index) with lambda_max = 2 is built")
        squeeze = x.dim() == 2
        if squeeze:
            x = x.unsqueeze(0)
        op = self._laplacian(edge_index, x.size(-2))
        weight = torch.stack([lin.weight.t() for lin in self.lins])        # [K, C_in, C_out]
        cin = self.in_channels
        wide = next((w for w in _LDS_WIDTHS if w >= cin), cin)
        if wide != cin and len(self.lins) > 1:
            # zero channels (and zero weight rows) up to the next kernel width: same sums, fast path
            x = F.pad(x, (0, wide - cin))
            weight = F.pad(weight, (0, 0, 0, wide - cin))
        out = F_hip.cheb_conv(x, weight, self.bias, op, relu=relu)
        return out.squeeze(0) if squeeze else out

    def __repr__(self):
        return '{}({}, {}, K={}, normalization={})'.format(
            self.__class__.__name__, self.in_channels, self.out_channels, len(self.lins), self.normalization)


class cheb_GCN(torch.nn.Module):

    def __init__(self, num_feature, config, downsample_matrices, upsample_matrices, adjacency_matrices, num_nodes):
        super().__init__()
        self.n_layers = config['n_layers']
        # the reference aliases and mutates the caller's list (cheb_cls.py:60-61; the reason crecon.py:241
        # re-reads its config) -- kept, callers may rely on seeing the widened list
        self.filters = config['num_conv_filters']
        self.filters.insert(0, num_feature)
        self.z = config['num_classes']
        self.K = config['polygon_order']
        self.downsample_matrices = downsample_matrices
        self.upsample_matrices = upsample_matrices
        self.adjacency_matrices = adjacency_matrices
        self.A_edge_index = []
        for i in range(len(num_nodes)):
            idx = self.adjacency_matrices[i]._indices()
            self.A_edge_index.append(idx[:, idx[0] != idx[1]])          # remove_self_loops (cheb_cls.py:70-72)
        f = self.filters
        self.cheb = torch.nn.ModuleList([ChebConv(f[i], f[i + 1], self.K[i]) for i in range(len(f) - 2)])
        self.enc_lin = torch.nn.Linear(self.downsample_matrices[-1].shape[0] * f[-2], 128)
        self.cls_layer = torch.nn.Linear(128, self.z)
        self.reset_parameters()

    def forward(self, data):
        x = data
        batch_size = x.shape[0]
        x = x.reshape(batch_size, -1, self.filters[0])
        for i in range(self.n_layers):
            x = self.cheb[i](x, self.A_edge_index[i], relu=True)       # conv + F.relu (cheb_cls.py:97-99), fused
            x = Pool(x, self.downsample_matrices[i])
        x = x.reshape(x.shape[0], self.enc_lin.in_features)
        x = F_hip.linear(x, self.enc_lin.weight, self.enc_lin.bias, relu=True)
        return F_hip.linear(x, self.cls_layer.weight, self.cls_layer.bias)

    def reset_parameters(self):
        torch.nn.init.normal_(self.enc_lin.weight, 0, 0.1)
        torch.nn.init.normal_(self.cls_layer.weight, 0, 0.1)
        for i in range(self.n_layers):
            self.cheb[i].reset_parameters()
        print('Reset parameters...')
