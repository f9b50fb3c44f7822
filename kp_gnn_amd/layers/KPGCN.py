"""KP-GCN convolution on the MI355X hot path.

Drop-in for the reference's layers/KPGCN.py `KPGCNConv` (:28-126).  The reference appends N self-loop
edges with code 1, computes a per-hop degree by scatter_add and a per-(edge,hop) norm tensor (:85-109);
here degree = CSR segment length + 1, the self loop is a closed-form term of the kernel epilogue and
`relu(sum norm*(x_j+e)) + peripheral` (+ the geometric combine) is one HIP launch."""
import torch
import torch.nn as nn

from .._lib import MODE_GCN
from ..ops import khop_aggregate
from ._base import EdgeCodeTables, KHopMessagePassing
from .combine import GeometricCombine, make_combine


class KPGCNConv(KHopMessagePassing, EdgeCodeTables):
    def __init__(self, input_size, output_size, K, num_hop1_edge=1, num_pe=1, combine="geometric"):
        super().__init__()
        assert output_size % K == 0
        self.K = K
        self.output_size = output_size
        self.output_dk = output_size // K
        self.hop_proj = nn.Linear(input_size, output_size)
        self._make_tables(self.output_dk, K, num_hop1_edge, num_pe)
        if K > 1:
            self.combine_proj = nn.Linear(self.output_dk, output_size)
            self.combine = make_combine(combine, K, self.output_dk)
        else:
            self.combine = torch.squeeze
            self.combine_proj = nn.Identity()
        self.reset_parameters()

    def reset_parameters(self):
        self._reset_tables()
        self.hop_proj.reset_parameters()
        if self.K > 1:
            self.combine.reset_parameters()
            self.combine_proj.reset_parameters()

    def forward(self, x, edge_index, edge_attr, pe_attr=None, peripheral_attr=None):
        n = x.size(0)
        csr, k_act = self._csr(edge_index, edge_attr, n)
        x = self.hop_proj(x).view(n, self.K, self.output_dk)
        x, xbias = self._path_encoding(x, pe_attr)
        t0, tk = self._tables()
        if isinstance(self.combine, GeometricCombine):
            h = khop_aggregate(x, csr, k_act, MODE_GCN, table0=t0, tablek=tk, periph=peripheral_attr,
                               theta=self.combine.alphas, xbias=xbias)
        else:
            h = self.combine(khop_aggregate(x, csr, k_act, MODE_GCN, table0=t0, tablek=tk, periph=peripheral_attr,
                                            xbias=xbias))
        return self.combine_proj(h)
