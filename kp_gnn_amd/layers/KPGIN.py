"""KP-GIN convolution on the MI355X hot path.

Drop-in for the reference's layers/KPGIN.py `KPGINConv` (:12-121): same constructor signature,
attributes (.K, .output_size, .output_dk, .input_dk), reset_parameters() and state_dict keys.
forward() maps to ONE fused HIP launch for   S + peripheral + (1+eps) x   (reference :90-105: edge
embeddings, propagate/message/aggregate, peripheral add, eps term), then the per-hop 2-layer MLP
(:106-109), the hop combine and combine_proj (:112)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .._lib import MODE_GIN
from ..ops import khop_aggregate
from ..ops_dense import hop_mlp, hop_mlp_supported
from ._base import EdgeCodeTables, KHopMessagePassing
from .combine import GeometricCombine, make_combine


class KPGINConv(KHopMessagePassing, EdgeCodeTables):
    def __init__(self, input_size, output_size, K, eps=0., train_eps=False, num_hop1_edge=1, num_pe=1,
                 combine="geometric"):
        super().__init__()
        assert input_size % K == 0
        assert output_size % K == 0
        self.K = K
        self.output_size = output_size
        self.input_dk = input_size // K
        self.output_dk = output_size // K
        dk_in, dk_out = self.input_dk, self.output_dk
        self.hop_proj1 = nn.Parameter(torch.empty(K, dk_in, dk_out))
        self.hop_bias1 = nn.Parameter(torch.empty(K, dk_out))
        self.hop_proj2 = nn.Parameter(torch.empty(K, dk_out, dk_out))
        self.hop_bias2 = nn.Parameter(torch.empty(K, dk_out))
        self.initial_eps = eps
        if train_eps:
            self.eps = nn.Parameter(torch.tensor([float(eps)]))
        else:
            self.register_buffer("eps", torch.tensor([float(eps)]))
        self._make_tables(dk_in, K, num_hop1_edge, num_pe)
        if K > 1:
            self.combine_proj = nn.Linear(dk_out, output_size)
            self.combine = make_combine(combine, K, dk_out)
        else:
            self.combine = torch.squeeze
            self.combine_proj = nn.Identity()
        self._fused_mlp = None
        self.reset_parameters()

    def reset_parameters(self):
        self._reset_tables()
        for w, b in ((self.hop_proj1, self.hop_bias1), (self.hop_proj2, self.hop_bias2)):
            nn.init.kaiming_uniform_(w)
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(w)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(b, -bound, bound)
        if self.K > 1:
            self.combine.reset_parameters()
            self.combine_proj.reset_parameters()
        nn.init.zeros_(self.eps)  # (the reference zeroes eps here whatever `eps` was passed, KPGIN.py:84)

    def forward(self, x, edge_index, edge_attr, pe_attr=None, peripheral_attr=None):
        n = x.size(0)
        state = x if (x.dim() == 2 and x.is_contiguous()) else None      # (the body's [N,H] state this layer reads)
        x = x.reshape(n, self.K, self.input_dk)
        x3 = x
        csr, k_act = self._csr(edge_index, edge_attr, n)
        x, xbias = self._path_encoding(x, pe_attr)
        t0, tk = self._tables()
        # (x_state: only when the aggregation reads the state itself - a live path encoding makes x a new tensor - and the
        #  caller marked this layer as the state's last reader, see body.py)
        share = state is not None and x is x3 and getattr(state, "_kp_last_reader", None) is self
        s = khop_aggregate(x, csr, k_act, MODE_GIN, table0=t0, tablek=tk, periph=peripheral_attr, eps=self.eps,
                           xbias=xbias, x_state=state if share else None)   # N,K,dk = x_n + P + (1+eps) x
        if self._fused_mlp is None:
            geo = isinstance(self.combine, GeometricCombine)
            self._fused_mlp = 2 if (geo and hop_mlp_supported(self.K, self.input_dk, self.output_dk, self.output_size)) \
                else (1 if hop_mlp_supported(self.K, self.input_dk, self.output_dk) else 0)
        mlp = (self.hop_proj1, self.hop_bias1, self.hop_proj2, self.hop_bias2)
        if self._fused_mlp == 2:  # per-hop MLP (:106-109), geometric combine and combine_proj (:112): one launch per direction
            return hop_mlp(s, *mlp, theta=self.combine.theta(), wc=self.combine_proj.weight, bc=self.combine_proj.bias)
        if self._fused_mlp == 1:
            if isinstance(self.combine, GeometricCombine):
                return self.combine_proj(hop_mlp(s, *mlp, theta=self.combine.theta()))
            return self.combine_proj(self.combine(hop_mlp(s, *mlp)))
        h = s.transpose(0, 1)                                           # K,N,dk  (hops wider than 32: BLAS)
        h = F.relu(torch.baddbmm(self.hop_bias1.unsqueeze(1), h, self.hop_proj1))
        h = F.relu(torch.baddbmm(self.hop_bias2.unsqueeze(1), h, self.hop_proj2))
        return self.combine_proj(self.combine(h.transpose(0, 1)))
