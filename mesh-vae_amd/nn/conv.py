"""`nn.conv` -- MI355X-native mirror of the reference module of the same name.

Exports `ChebConv_batch` with the reference's constructor, attributes, `norm` static
method and `forward(x, edge_index, norm)` signature (reference nn/conv.py:532-581); the
arithmetic runs in hand-written HIP kernels (libmeshvae_hip: mvh_cheb_conv_fwd/_bwd) instead
of the reference's materialised-message MessagePassing (nn/conv.py:80-385).  The other
convolutions in the reference file (ChebConv, Spatial_conv, graph_attention) are never
instantiated by cheb_VAE and are out of scope.
"""
import torch
from torch.nn import Parameter

from meshvae_hip import functional as F_hip
from meshvae_hip import topology


class ChebConv_batch(torch.nn.Module):
    """Batched Chebyshev spectral convolution on a fixed topology.

    out = sum_k T_k(L) x W_k + b with T_0 = x, T_1 = L x, T_k = 2 L T_{k-1} - T_{k-2},
    where L is given by (edge_index, norm) (reference nn/conv.py:557-577).
    """

    def __init__(self, in_channels, out_channels, K, normalization=None, bias=True):
        super().__init__()
        assert K > 0                                        # nn/conv.py:445
        assert normalization in [None, 'sym', 'rw'], 'Invalid normalization'
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.normalization = normalization
        self.weight = Parameter(torch.Tensor(K, in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        # N(0, 0.1) for weight then bias: same RNG order as nn/conv.py:536-538
        torch.nn.init.normal_(self.weight, mean=0, std=0.1)
        if self.bias is not None:
            torch.nn.init.normal_(self.bias, mean=0, std=0.1)

    @staticmethod
    def norm(edge_index, num_nodes, edge_weight=None, dtype=None):
        """(edge_index, -deg^-1/2[row] * w * deg^-1/2[col]) after dropping self loops
        (reference nn/conv.py:541-555).  Init-time only."""
        dev = edge_index.device
        # init-time, O(E) host work: done on the CPU so the values are bit-identical to the
        # reference's CPU path regardless of the device's pow() rounding, then moved back
        ei = edge_index.cpu()
        keep = ei[0] != ei[1]
        ei = ei[:, keep]
        if edge_weight is None:
            w = torch.ones((ei.size(1),), dtype=dtype)
        else:
            w = edge_weight.cpu()[keep]
        row, col = ei
        deg = torch.zeros(num_nodes, dtype=w.dtype).scatter_add_(0, row, w)
        deg_inv_sqrt = deg.pow(-0.5)
        deg_inv_sqrt[deg_inv_sqrt == float('inf')] = 0
        return ei.to(dev), (-deg_inv_sqrt[row] * w * deg_inv_sqrt[col]).to(dev)

    def forward(self, x, edge_index, norm, edge_weight=None, relu=False):
        """x [B, N, C_in], edge_index [2, E] int64, norm [E] -> [B, N, C_out].
        `relu=True` fuses the F.relu the model applies right after (cheb_VAE.py:264,285)."""
        op = topology.laplacian(edge_index, norm, x.size(1))
        return F_hip.cheb_conv(x, self.weight, self.bias, op, relu=relu)

    def __repr__(self):
        return '{}({}, {}, K={}, normalization={})'.format(
            self.__class__.__name__, self.in_channels, self.out_channels,
            self.weight.size(0), self.normalization)
