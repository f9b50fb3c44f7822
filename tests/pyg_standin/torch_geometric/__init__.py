"""TEST-ONLY stand-in for the handful of torch_geometric symbols the KP-GNN reference touches.

PyG (pin 2.1.0, reference README.md:12) is a third-party dependency of the reference that is not
installed in this image and cannot be fetched.  This package restates PyG's *documented* contract for
those symbols so that the reference's own files can be imported in the build container by
tests/golden/make_golden.py to produce golden vectors.  It is our code (it restates PyG, not the
reference), it is never imported by the product package, and nothing here travels as "reference".

Semantics restated:
  * nn.MessagePassing(aggr='add', node_dim=0).propagate(edge_index, **kw): for every parameter p of
    self.message: p ending in '_j' -> kw[p[:-2]].index_select(0, edge_index[0]); '_i' ->
    index_select(0, edge_index[1]); otherwise kw[p] unchanged.  m = message(...);
    out = zeros([N, *m.shape[1:]]).index_add_(0, edge_index[1], m); return update(out).
  * utils.add_self_loops(ei, num_nodes=N) appends arange(N) twice and returns (ei', None).
  * utils.to_scipy_sparse_matrix(ei, attr=None, num_nodes) -> scipy.sparse.coo_matrix.
  * data.Data: attribute bag, `in` == "present and not None", num_nodes = x.size(0).
"""
__version__ = "0.0-standin"
