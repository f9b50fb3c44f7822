"""Mask-only K-hop GIN layer of the reference's regular-graph simulation (run_simulation.py `KGINConv`
:29-93): no edge-code embeddings, no peripheral features, concat-combine.  Same fused aggregation kernel with
`use_tables = 0`.  The reference reads a module-global `args.graph` to decide on sum pooling (:83-84); here
that is the constructor flag `pool`."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .._lib import MODE_GIN
from ..khop_csr import get_khop_csr
from ..ops import khop_aggregate
from ._base import KHopMessagePassing


class KGINConv(KHopMessagePassing):
    def __init__(self, hidden_size, K, eps=0., train_eps=False, pool=False):
        super().__init__()
        self.K = K
        self.hidden_size = hidden_size
        self.pool = pool
        self.proj = nn.Linear(1, K * hidden_size)
        self.hop_proj1 = nn.Parameter(torch.empty(K, hidden_size, hidden_size))
        self.hop_bias1 = nn.Parameter(torch.empty(K, hidden_size))
        self.hop_proj2 = nn.Parameter(torch.empty(K, hidden_size, hidden_size))
        self.hop_bias2 = nn.Parameter(torch.empty(K, hidden_size))
        self.initial_eps = eps
        if train_eps:
            self.eps = nn.Parameter(torch.tensor([float(eps)]))
        else:
            self.register_buffer("eps", torch.tensor([float(eps)]))
        self.combine_proj = nn.Linear(hidden_size * K, hidden_size)
        self.reset_parameters()

    def reset_parameters(self):
        for w, b in ((self.hop_proj1, self.hop_bias1), (self.hop_proj2, self.hop_bias2)):
            nn.init.kaiming_uniform_(w)
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(w)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(b, -bound, bound)
        self.combine_proj.reset_parameters()
        nn.init.zeros_(self.eps)

    def forward(self, x, edge_index, edge_attr, batch=None):
        n = x.size(0)
        x = self.proj(x).view(n, self.K, self.hidden_size)
        csr, k_act = get_khop_csr(edge_index, edge_attr, n)
        s = khop_aggregate(x, csr, k_act, MODE_GIN, eps=self.eps)             # x_n + (1+eps) x, mask only
        from ..ops_dense import hop_mlp, hop_mlp_supported
        if s.is_cuda and hop_mlp_supported(self.K, self.hidden_size, self.hidden_size):
            # the per-hop 2-layer MLP (run_simulation.py:76-79) on the MFMA kernel the KP-GIN layers use: one launch
            h2 = hop_mlp(s, self.hop_proj1, self.hop_bias1, self.hop_proj2, self.hop_bias2)      # N,K,h
            out = self.combine_proj(h2.reshape(n, self.K * self.hidden_size))
        else:
            h = s.transpose(0, 1)
            h = F.relu(torch.baddbmm(self.hop_bias1.unsqueeze(1), h, self.hop_proj1))
            h = F.relu(torch.baddbmm(self.hop_bias2.unsqueeze(1), h, self.hop_proj2))
            out = self.combine_proj(h.transpose(0, 1).reshape(n, self.K * self.hidden_size))
        if self.pool:
            from ..ops import segment_pool
            out = segment_pool(out, batch, int(batch[-1].item()) + 1)     # global_add_pool (run_simulation.py:83-84)
        return out
