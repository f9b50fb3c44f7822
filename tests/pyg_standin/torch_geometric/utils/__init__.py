"""Stand-in for torch_geometric.utils (test-only; see package docstring)."""
import scipy.sparse
import torch


def add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    n = int(edge_index.max().item()) + 1 if num_nodes is None else num_nodes
    loop = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
    loop = loop.unsqueeze(0).repeat(2, 1)
    return torch.cat([edge_index, loop], dim=1), None


def to_scipy_sparse_matrix(edge_index, edge_attr=None, num_nodes=None):
    row, col = edge_index.cpu()
    if edge_attr is None:
        edge_attr = torch.ones(row.size(0))
    else:
        edge_attr = edge_attr.view(-1).cpu()
        assert edge_attr.size(0) == row.size(0)
    n = int(edge_index.max().item()) + 1 if num_nodes is None else num_nodes
    return scipy.sparse.coo_matrix((edge_attr.numpy(), (row.numpy(), col.numpy())), (n, n))
