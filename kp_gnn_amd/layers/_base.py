"""Shared plumbing of the K-hop conv layers (not part of the reference's surface)."""
import torch
import torch.nn as nn
from ..khop_csr import get_khop_csr, path_encoding_is_zero
from ..ops import embedding_rows

try:  # the reference's layers subclass PyG's MessagePassing (layers/KPGIN.py:12); keep that when PyG exists
    from torch_geometric.nn import MessagePassing as _PyGMessagePassing

    class KHopMessagePassing(_PyGMessagePassing):
        def __init__(self):
            super().__init__(node_dim=0)
            self.aggr = "add"
except Exception:  # PyG is not installed in this image / on the GPU box

    class KHopMessagePassing(nn.Module):
        """Stand-alone base: the HIP operator replaces propagate()/message()/update(), which the
        reference's callers (models/GNNs.py:190,429,655,679) never invoke directly."""

        def __init__(self):
            super().__init__()
            self.aggr = "add"
            self.node_dim = 0


class EdgeCodeTables(object):
    """Mixin: builds hop1_edge_emb / hopk_edge_emb / hopk_node_path_emb exactly as the reference's
    constructors do (KPGIN.py:48-53, KPGINplus.py:32-35, KPGCN.py:51-55): +2 rows for mask(0) and
    self-loop(1), padding_idx=0."""

    def _make_tables(self, width, K, num_hop1_edge, num_pe):
        self.hop1_edge_emb = nn.Embedding(num_hop1_edge + 2, width, padding_idx=0)
        if K > 1:
            self.hopk_edge_emb = nn.Embedding(num_pe + 2, width, padding_idx=0)
            self.hopk_node_path_emb = nn.Embedding(num_pe, width, padding_idx=0)
        else:
            self.hopk_edge_emb = None

    def _reset_tables(self):
        self.hop1_edge_emb.reset_parameters()
        if self.K > 1:
            self.hopk_edge_emb.reset_parameters()
            self.hopk_node_path_emb.reset_parameters()

    def _tables(self):
        return self.hop1_edge_emb.weight, (self.hopk_edge_emb.weight if self.K > 1 else None)

    def _csr(self, edge_index, edge_attr, num_nodes):
        return get_khop_csr(edge_index, edge_attr, num_nodes)

    def _path_encoding(self, x, pe_attr):
        """`x[:, 1:] += hopk_node_path_emb(pe_attr)` of the reference (KPGIN.py:92-94) without mutating
        the caller's tensor.  Returns (x, xbias): when pe_attr is all padding (always, for the
        reference's own pre-transform) the add degenerates to one constant row that the kernel folds
        into its epilogue, so no [N,K,D] pass is spent on adding zeros."""
        if self.K == 1 or pe_attr is None:
            return x, None
        if isinstance(x, list) or path_encoding_is_zero(pe_attr):  # (lists only arrive with all-padding pe_attr)
            return x, self.hopk_node_path_emb.weight[0].detach()
        # (non-zero path codes never come out of the reference's own pre-transform, Q1; the lookup goes through the
        #  gather-sum kernels so that this path stays hipGraph-capturable - torch's embedding backward is not)
        pe = embedding_rows(self.hopk_node_path_emb.weight, pe_attr, padding_idx=0)
        return torch.cat([x[:, :1], x[:, 1:] + pe], dim=1), None
