"""`nn.pool` -- MI355X-native mirror of the reference's SurfacePool (nn/pool.py:13-23).

`SurfacePool().forward(x, pool_mat)` computes P @ x per mesh for a torch sparse COO matrix
P [N_out, N_in] (D: one-hot rows; U: 3 barycentric taps per row) with a hand-written HIP
gather kernel (libmeshvae_hip: mvh_pool_fwd/_bwd), bit-exact w.r.t. the reference's
index_select -> mul -> scatter_add_.  SortPool / DIFFPool of the reference file are unused
by cheb_VAE (DIFFPool cannot even be constructed) and are out of scope.
"""
import torch

from meshvae_hip import functional as F_hip
from meshvae_hip import topology


class SurfacePool(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.flow = 'target_to_source'

    def forward(self, x, pool_mat, dtype=None):
        return F_hip.surface_pool(x, topology.pool_operator(pool_mat))
