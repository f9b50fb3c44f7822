"""Stand-in for torch_geometric.nn (test-only; see package docstring)."""
import inspect

import torch
import torch.nn as nn


class MessagePassing(nn.Module):
    """PyG MessagePassing contract, sum aggregation, flow source_to_target, node_dim=0."""

    def __init__(self, aggr="add", node_dim=0, **kwargs):
        super().__init__()
        self.aggr = aggr
        self.node_dim = node_dim
        assert node_dim == 0

    def propagate(self, edge_index, size=None, **kwargs):
        params = list(inspect.signature(self.message).parameters)
        args = []
        for p in params:
            if p.endswith("_j"):
                args.append(kwargs[p[:-2]].index_select(0, edge_index[0]))
            elif p.endswith("_i"):
                args.append(kwargs[p[:-2]].index_select(0, edge_index[1]))
            else:
                args.append(kwargs[p])
        msg = self.message(*args)
        num_nodes = kwargs["x"].size(0)
        out = torch.zeros([num_nodes] + list(msg.shape[1:]), dtype=msg.dtype, device=msg.device)
        out = out.index_add_(0, edge_index[1], msg)
        return self.update(out)

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out


def global_add_pool(x, batch, size=None):
    size = int(batch.max().item()) + 1 if size is None else size
    out = torch.zeros([size] + list(x.shape[1:]), dtype=x.dtype, device=x.device)
    return out.index_add_(0, batch, x)


def global_mean_pool(x, batch, size=None):
    size = int(batch.max().item()) + 1 if size is None else size
    cnt = torch.zeros(size, dtype=x.dtype, device=x.device).index_add_(0, batch, torch.ones_like(batch, dtype=x.dtype))
    return global_add_pool(x, batch, size) / cnt.clamp(min=1).unsqueeze(-1)


def global_max_pool(x, batch, size=None):
    size = int(batch.max().item()) + 1 if size is None else size
    out = torch.full([size] + list(x.shape[1:]), float("-inf"), dtype=x.dtype, device=x.device)
    idx = batch.view(-1, *([1] * (x.dim() - 1))).expand_as(x)
    return out.scatter_reduce(0, idx, x, reduce="amax")


class BatchNorm(nn.Module):
    """PyG BatchNorm = thin wrapper over nn.BatchNorm1d with attribute name `module`."""

    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.module = nn.BatchNorm1d(in_channels, eps, momentum, affine, track_running_stats)

    def reset_parameters(self):
        self.module.reset_parameters()

    def forward(self, x):
        return self.module(x)


class _Unsupported(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, *a, **k):
        raise NotImplementedError("stand-in: only BatchNorm is restated")


class LayerNorm(_Unsupported):
    pass


class InstanceNorm(_Unsupported):
    pass


class PairNorm(_Unsupported):
    pass


class GraphSizeNorm(_Unsupported):
    pass


class AttentionalAggregation(_Unsupported):
    def __init__(self, gate_nn=None, nn=None):
        super().__init__()


class DataParallel(_Unsupported):
    pass
