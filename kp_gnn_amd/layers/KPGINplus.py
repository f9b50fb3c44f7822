"""KP-GIN+ convolution on the MI355X hot path.

Drop-in for the reference's layers/KPGINplus.py `KPGINPlusConv` (:11-88): x is the [N,k,H] stack of
the previous layers' states; same constructor, attributes and state_dict keys.  With the geometric
combine the whole of  combine( gelu(S) + peripheral )  (:64-78, combine.py:43-46) is ONE HIP launch
that writes [N,H] directly; only the Linear-BN-ReLU x2 MLP (:25-30) is left to library GEMMs."""
import torch
import torch.nn as nn

from .._lib import MODE_GINPLUS
from ..ops import khop_aggregate
from ..ops_dense import mlp_linear_bn_relu_x2
from ._base import EdgeCodeTables, KHopMessagePassing
from .combine import GeometricCombine, make_combine


class KPGINPlusConv(KHopMessagePassing, EdgeCodeTables):
    def __init__(self, input_size, output_size, K, num_hop1_edge=1, num_pe=1, combine="independent"):
        super().__init__()
        self.K = K
        self.output_size = output_size
        self.mlp = nn.Sequential(nn.Linear(input_size, output_size), nn.BatchNorm1d(output_size), nn.ReLU(),
                                 nn.Linear(output_size, output_size), nn.BatchNorm1d(output_size), nn.ReLU())
        self._make_tables(input_size, K, num_hop1_edge, num_pe)
        if K > 1:
            self.combine = make_combine(combine, K, output_size)  # raises for "independent" like the reference
        else:
            self.combine = torch.squeeze
            # (one hop: the combine is the identity; on the GPU the layer asks for the fused epilogue with theta = 1, so that its
            #  backward is the fused combine + table-gradient kernel like every other layer's: 3 launches instead of 6)
            self.register_buffer("_theta_one", torch.ones(1, input_size), persistent=False)
        self.reset_parameters()

    def reset_parameters(self):
        self._reset_tables()
        for m in self.mlp:
            if hasattr(m, "reset_parameters"):
                m.reset_parameters()
        if self.K > 1:
            self.combine.reset_parameters()

    def forward_slots(self, h_slots, edge_index, edge_attr, pe_attr=None, peripheral_attr=None, history=False, post_norm=None):
        """Same as forward(torch.stack(h_slots, 1), ...) without the stacking copy (and without the slicing
        adds in backward): hop slot m reads h_slots[m] ([N,H]) where it lives.  Extension of the reference API
        used by kp_gnn_amd.body.GNNPlus; GNNs.py:413-418 builds the stack with torch.cat every layer.
        post_norm = (nn.BatchNorm1d, residual or None): return bn(result) + residual instead (the body's per-layer norm,
        GNNs.py:440-441, taken into the layer's last autograd node: ops_dense.FusedMLP)."""
        from ..khop_csr import path_encoding_is_zero
        if self.K > 1 and pe_attr is not None and not path_encoding_is_zero(pe_attr):
            return self.forward(torch.stack(list(h_slots), dim=1), edge_index, edge_attr, pe_attr, peripheral_attr,
                                _post_norm=post_norm)
        return self.forward(list(h_slots), edge_index, edge_attr, pe_attr, peripheral_attr, _history=history, _post_norm=post_norm)

    def forward(self, x, edge_index, edge_attr, pe_attr=None, peripheral_attr=None, _history=False, _post_norm=None):
        n = x[0].size(0) if isinstance(x, list) else x.size(0)
        csr, k_act = self._csr(edge_index, edge_attr, n)
        x, xbias = self._path_encoding(x, pe_attr)
        t0, tk = self._tables()
        if isinstance(self.combine, GeometricCombine):
            h = khop_aggregate(x, csr, k_act, MODE_GINPLUS, table0=t0, tablek=tk, periph=peripheral_attr,
                               theta=self.combine.alphas, xbias=xbias, share_slot_grads=_history)   # N,H
        elif self.K == 1 and k_act == 1 and self._theta_one.is_cuda and n > 1:
            h = khop_aggregate(x, csr, k_act, MODE_GINPLUS, table0=t0, tablek=tk, periph=peripheral_attr,
                               theta=self._theta_one, xbias=xbias, share_slot_grads=_history)       # N,H (= squeeze of N,1,H)
        else:
            xn = khop_aggregate(x, csr, k_act, MODE_GINPLUS, table0=t0, tablek=tk, periph=peripheral_attr,
                                xbias=xbias, share_slot_grads=_history)                       # N,k,H
            h = self.combine(xn)
        # (_kp_emit_out_stats: set by a caller that applies a BatchNorm to the result next - kp_gnn_amd.body does)
        return mlp_linear_bn_relu_x2(self.mlp, h, emit_out_stats=getattr(self, "_kp_emit_out_stats", False), post_norm=_post_norm)
