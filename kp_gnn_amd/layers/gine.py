"""GINE convolution over the K-hop edge list (used by KP-GIN' for layers 2..L).

Drop-in for the reference's layers/gine.py `GINEConv` (:9-59): it is handed the WHOLE K-hop edge list
with only the hop-1 column of edge_attr (models/GNNs.py:679), i.e. ~90 % masked rows at K=16; here
that is simply the k=1 prefix of the batch's K-hop CSR."""
import torch
import torch.nn as nn

from .._lib import MODE_GIN
from ..khop_csr import get_khop_csr
from ..ops import khop_aggregate
from ..ops_dense import mlp_linear_bn_relu_x2
from ._base import KHopMessagePassing


class GINEConv(KHopMessagePassing):
    def __init__(self, input_size, output_size, eps=0., num_hop1_edge=1, train_eps=False):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.initial_eps = eps
        if train_eps:
            self.eps = nn.Parameter(torch.tensor([float(eps)]))
        else:
            self.register_buffer("eps", torch.tensor([float(eps)]))
        self.mlp = nn.Sequential(nn.Linear(input_size, output_size), nn.BatchNorm1d(output_size), nn.ReLU(),
                                 nn.Linear(output_size, output_size), nn.BatchNorm1d(output_size), nn.ReLU())
        self.hop1_edge_emb = nn.Embedding(num_hop1_edge + 2, input_size, padding_idx=0)
        self.reset_parameters()

    def reset_parameters(self):
        for m in self.mlp:
            if hasattr(m, "reset_parameters"):
                m.reset_parameters()
        self.hop1_edge_emb.reset_parameters()
        self.eps.data.fill_(self.initial_eps)

    def forward(self, x, edge_index, edge_attr):
        n = x.size(0)
        csr, k_act = get_khop_csr(edge_index, edge_attr, n)
        # (x_state: the body marked this layer as the LAST of the state's readers to run backward - ops.khop_aggregate)
        share = x.dim() == 2 and x.is_contiguous() and getattr(x, "_kp_last_reader", None) is self
        out = khop_aggregate(x.reshape(n, 1, self.input_size), csr, k_act, MODE_GIN,
                             table0=self.hop1_edge_emb.weight, eps=self.eps, x_state=x if share else None)
        return mlp_linear_bn_relu_x2(self.mlp, out.squeeze(1), emit_out_stats=getattr(self, "_kp_emit_out_stats", False))
