"""Collated K-hop batch: what PyG's DataLoader + the reference's pre_transform hand to the model
(train_ZINC.py:209-224), as one plain attribute bag whose tensors live in HBM, with the K-hop CSR built
once and attached so that no layer call pays for it."""
import torch

from . import khop_transform as KT
from .khop_csr import KHopCSR, attach_khop_csr

_TENSORS = ("x", "z", "edge_index", "edge_attr", "pe_attr", "peripheral_edge_attr", "peripheral_configuration_attr",
            "batch", "y")


class KHopBatch:
    def __init__(self, **kw):
        self.num_graphs = kw.pop("num_graphs", None)
        self.csr = None
        for k in _TENSORS:
            setattr(self, k, kw.get(k))

    @property
    def num_nodes(self):
        return self.x.size(0)

    def to(self, device):
        out = KHopBatch(num_graphs=self.num_graphs)
        for k in _TENSORS:
            v = getattr(self, k)
            setattr(out, k, None if v is None else v.to(device, non_blocking=True))
        return out

    def build_csr(self):
        """Build the device CSR now and pin it to this batch's edge_index (layers pick it up)."""
        self.csr = KHopCSR.build(self.edge_index, self.edge_attr, self.num_nodes)
        if self.batch is not None and self.num_graphs:
            from .ops import graph_ptr_of
            gp = graph_ptr_of(self.batch, self.num_graphs)
            self.csr.graph_ptr = gp
            self.csr.max_graph_nodes = int((gp[1:] - gp[:-1]).max().item()) if self.num_graphs > 0 else 0
        attach_khop_csr(self.edge_index, self.edge_attr, self.num_nodes, self.csr)
        return self.csr

    def as_dict(self):
        return {k: getattr(self, k) for k in _TENSORS if getattr(self, k) is not None}


def collate_khop(node_ptr, edge_ptr, edge_index, edge_attr, x, khop_args, y=None, num_threads=0):
    """Raw graphs (CSR-style concatenation, local node ids) -> exact K-hop pre-transform -> collated batch."""
    out = KT.khop_batch(node_ptr, edge_ptr, edge_index, edge_attr, *khop_args, num_threads=num_threads)
    xt = torch.as_tensor(x)
    if xt.dim() == 1:
        xt = xt.view(-1, 1)
    b = KHopBatch(x=xt, edge_index=out["edge_index"], edge_attr=out["edge_attr"], pe_attr=out["pe_attr"],
                  peripheral_edge_attr=out["peripheral_edge_attr"],
                  peripheral_configuration_attr=out["peripheral_configuration_attr"], batch=out["batch"], y=y,
                  num_graphs=len(node_ptr) - 1)
    b.node_ptr = out["node_ptr"].numpy()          # per-graph slices (PyG's slices['x']): what KHopDataset.from_collated needs
    b.edge_ptr = out["edge_ptr"].numpy()
    return b


def synthetic_zinc_batch(num_graphs, seed0, K=8, kernel="spd", num_threads=0):
    """ZINC-12k-shaped synthetic batch with the reference's ZINC pre-transform arguments
    (train_ZINC.py:125-134,191-194: max_pe_num=50, max_hop_num=6, max_edge_type=3, counts 50)."""
    node_ptr, edge_ptr, ei, ea, x = KT.synth_molecules(num_graphs, seed0)
    g = torch.Generator().manual_seed(int(seed0))
    y = torch.randn(num_graphs, generator=g)
    return collate_khop(node_ptr, edge_ptr, ei, ea, x, (K, 50, 6, 3, 50, 50, kernel), y=y, num_threads=num_threads)


def synthetic_qm9_batch(num_graphs, seed0, K=6, kernel="spd", num_threads=0):
    """QM9-shaped synthetic batch with the reference's QM9 pre-transform arguments (train_qm9.py:141-158: max_pe_num=50,
    max_hop_num=5, max_edge_type=4, max_edge_count=20, max_distance_count=15): 11 node features (PyG QM9's x) + the atomic
    number z, whose 8-wide embedding makes up the encoder's 19 inputs (layers/input_encoder.py:43-84), one regression target."""
    node_ptr, edge_ptr, ei, ea, atom = KT.synth_molecules(num_graphs, seed0, shape=KT.qm9_shape())
    g = torch.Generator().manual_seed(int(seed0))
    n = int(node_ptr[-1])
    x = torch.randn(n, 11, generator=g)
    z = torch.tensor([1, 6, 7, 8, 9])[torch.as_tensor(atom)]          # H, C, N, O, F
    y = torch.randn(num_graphs, generator=g)
    b = collate_khop(node_ptr, edge_ptr, ei, ea, x, (K, 50, 5, 4, 20, 15, kernel), y=y, num_threads=num_threads)
    b.z = z
    return b


def synthetic_regular_batch(num_graphs, seed0, n=1280, degree=3, K=8, num_threads=0):
    """run_simulation.py's workload (:96-129): `num_graphs` random `degree`-regular graphs on n nodes
    (nx.random_regular_graph(d, n, seed), seeds seed0, seed0+1, ..), x = ones [n,1], pre-transform arguments
    (k, 10, 1, 1, 1, 1, "spd") (:103)."""
    import networkx as nx
    import numpy as np
    eis, node_ptr, edge_ptr = [], [0], [0]
    for s in range(num_graphs):
        G = nx.random_regular_graph(d=degree, n=n, seed=int(seed0) + s)
        e = np.array(sorted(G.to_directed().edges), dtype=np.int64).T      # (src, dst) sorted, as from_networkx emits them
        eis.append(e)
        node_ptr.append(node_ptr[-1] + n)
        edge_ptr.append(edge_ptr[-1] + e.shape[1])
    ei = np.ascontiguousarray(np.concatenate(eis, axis=1))
    x = torch.ones(num_graphs * n, 1)
    return collate_khop(np.array(node_ptr, dtype=np.int64), np.array(edge_ptr, dtype=np.int64), ei, None, x,
                        (K, 10, 1, 1, 1, 1, "spd"), num_threads=num_threads)
