"""KP-GraphSAGE convolution on the MI355X hot path.

Drop-in for the reference's layers/KPGraphSAGE.py `KPGraphSAGEConv` (:12-106).  The neighbour aggregation is
the same fused kernel as KP-GIN's (sum of x_j + edge-code rows over the active pairs, + peripheral); the
concat([x, x_n]) -> per-hop projection -> ReLU -> L2-normalise tail (:88-92) stays on library ops.
Note: the reference assigns `self.aggr` after MessagePassing.__init__, so PyG >= 2.1 keeps the
constructor's "add" reduction whatever `aggr` says; the scripts pass aggr="add" anyway (train_TU.py:328)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .._lib import MODE_SUM
from ..ops import khop_aggregate
from ._base import EdgeCodeTables, KHopMessagePassing
from .combine import make_combine


class KPGraphSAGEConv(KHopMessagePassing, EdgeCodeTables):
    def __init__(self, input_size, output_size, K, aggr="mean", num_hop1_edge=1, num_pe=1, combine="geometric"):
        super().__init__()
        self.aggr = aggr
        self.K = K
        assert input_size % K == 0
        assert output_size % K == 0
        self.input_dk = input_size // K
        self.output_dk = output_size // K
        self.output_size = output_size
        self.hop_proj = nn.Parameter(torch.empty(K, 2 * self.input_dk, self.output_dk))
        self.hop_bias = nn.Parameter(torch.empty(K, self.output_dk))
        self._make_tables(self.input_dk, K, num_hop1_edge, num_pe)
        if K > 1:
            self.combine_proj = nn.Linear(self.output_dk, output_size)
            self.combine = make_combine(combine, K, self.output_dk)
        else:
            self.combine = torch.squeeze
            self.combine_proj = nn.Identity()
        self.reset_parameters()

    def reset_parameters(self):
        self._reset_tables()
        if self.K > 1:
            self.combine.reset_parameters()
            self.combine_proj.reset_parameters()
        nn.init.kaiming_uniform_(self.hop_proj)
        fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.hop_proj)
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        nn.init.uniform_(self.hop_bias, -bound, bound)

    def forward(self, x, edge_index, edge_attr, pe_attr=None, peripheral_attr=None):
        n = x.size(0)
        x = x.reshape(n, self.K, self.input_dk)
        csr, k_act = self._csr(edge_index, edge_attr, n)
        x, xbias = self._path_encoding(x, pe_attr)
        t0, tk = self._tables()
        x_n = khop_aggregate(x, csr, k_act, MODE_SUM, table0=t0, tablek=tk, periph=peripheral_attr, xbias=xbias)
        if xbias is not None:  # the reference concatenates the path-encoded x (:88), not the raw input
            x = torch.cat([x[:, :1], x[:, 1:] + xbias], dim=1)
        h = torch.cat([x, x_n], dim=-1).transpose(0, 1)                       # K,N,2dk
        h = torch.baddbmm(self.hop_bias.unsqueeze(1), h, self.hop_proj).transpose(0, 1)
        h = F.normalize(F.relu(h), p=2, dim=-1)
        return self.combine_proj(self.combine(h))
