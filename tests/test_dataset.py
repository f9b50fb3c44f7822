"""Dataset-resident K-hop CSR + per-step collate (kp_gnn_amd/dataset.py, csrc/collate.hip) against the per-batch device
build on the PyG-style collated batch of the same graphs in the same order: every index array BIT-EXACT.

Reference behaviour being replaced: datasets/ZINC_dataset.py:139-140 (the (data, slices) store of the pre-transformed
dataset) + train_ZINC.py:224,36-40 (DataLoader(shuffle=True) -> Batch.from_data_list -> .to(device) every step)."""
import argparse

import numpy as np
import pytest
import torch

from kp_gnn_amd import khop_transform as KT
from kp_gnn_amd.batch import collate_khop

ZINC_ARGS = (8, 50, 6, 3, 50, 50, "spd")


class RawGraphs:
    """G raw graphs in concatenated form (local node ids), with a subset operator."""

    def __init__(self, node_ptr, edge_ptr, ei, ea, x, y):
        self.node_ptr, self.edge_ptr, self.ei, self.ea, self.x, self.y = node_ptr, edge_ptr, ei, ea, x, y

    @property
    def G(self):
        return len(self.node_ptr) - 1

    def subset(self, ids):
        nps, eps, eis, eas, xs = [0], [0], [], [], []
        for g in ids:
            n0, n1, e0, e1 = self.node_ptr[g], self.node_ptr[g + 1], self.edge_ptr[g], self.edge_ptr[g + 1]
            nps.append(nps[-1] + n1 - n0)
            eps.append(eps[-1] + e1 - e0)
            eis.append(self.ei[:, e0:e1])
            if self.ea is not None:
                eas.append(self.ea[e0:e1])
            xs.append(self.x[n0:n1])
        ea = np.concatenate(eas) if self.ea is not None else None
        return RawGraphs(np.array(nps, dtype=np.int64), np.array(eps, dtype=np.int64), np.ascontiguousarray(np.concatenate(eis, axis=1)),
                         ea, np.concatenate(xs), self.y[torch.as_tensor(np.asarray(ids, dtype=np.int64))])

    def collated(self, args):
        return collate_khop(self.node_ptr, self.edge_ptr, self.ei, self.ea, torch.as_tensor(self.x), args, y=self.y, num_threads=4)


def molecules(G, seed0=11):
    node_ptr, edge_ptr, ei, ea, x = KT.synth_molecules(G, seed0)
    return RawGraphs(node_ptr, edge_ptr, ei, ea, x, torch.randn(G, generator=torch.Generator().manual_seed(seed0)))


def odd_graphs():
    """A star with 200 leaves (runs of equal entries far beyond the 64 a list entry can count), a triangle, a path, a
    4-clique and a second star: multiplicity cuts, tiny graphs and tiles that straddle several graphs."""
    def sym(pairs):
        return [(a, b) for a, b in pairs] + [(b, a) for a, b in pairs]
    graphs = [
        (201, sym([(0, i) for i in range(1, 201)])),
        (3, sym([(0, 1), (1, 2), (0, 2)])),
        (6, sym([(i, i + 1) for i in range(5)])),
        (4, sym([(a, b) for a in range(4) for b in range(a + 1, 4)])),
        (2, sym([(0, 1)])),
        (90, sym([(0, i) for i in range(1, 90)])),
        (5, sym([(0, 1), (1, 2), (2, 3), (3, 4), (4, 0)])),
    ]
    nps, eps, eis, xs = [0], [0], [], []
    for n, e in graphs:
        e = sorted(e)
        nps.append(nps[-1] + n)
        eps.append(eps[-1] + len(e))
        eis.append(np.array(e, dtype=np.int64).T)
        xs.append(np.arange(n, dtype=np.int64) % 7)
    G = len(graphs)
    return RawGraphs(np.array(nps, dtype=np.int64), np.array(eps, dtype=np.int64), np.ascontiguousarray(np.concatenate(eis, axis=1)),
                     None, np.concatenate(xs), torch.arange(G, dtype=torch.float32))


def test_plan_header_is_the_three_prefix_sums():
    """Host side only (no GPU): the header kpgnn_collate reads."""
    from kp_gnn_amd.dataset import KHopDataset
    ds = KHopDataset()
    ds.G, ds.K = 5, 3
    ds.h_nodes = np.array([4, 0, 7, 2, 9])
    ds.h_pairs = np.array([10, 0, 30, 2, 50])
    ds.h_ents = np.array([3, 0, 5, 1, 8])
    hdr, B, N, A, n_ent = ds.plan([4, 0, 2, 2])
    assert (B, N, A, n_ent) == (4, 9 + 4 + 7 + 7, 50 + 10 + 30 + 30, 8 + 3 + 5 + 5)
    assert hdr.dtype == np.int32 and hdr.shape == (4 * 4 + 3,)
    assert hdr[:4].tolist() == [4, 0, 2, 2]
    assert hdr[4:9].tolist() == [0, 9, 13, 20, 27]
    assert hdr[9:14].tolist() == [0, 50, 60, 90, 120]
    assert hdr[14:19].tolist() == [0, 8, 11, 16, 21]
    with pytest.raises(IndexError):
        ds.plan([5])
    with pytest.raises(IndexError):
        ds.plan([])


def _check_equal_to_plain_build(ds, raw, ids, args):
    """collate(ids) against KHopCSR.build on the host-collated batch of the same graphs: bit for bit."""
    dev = torch.device("cuda:0")
    got = ds.collate(ids)
    ref = raw.subset(ids).collated(args).to(dev)
    ref.build_csr()
    g, r = got.csr, ref.csr
    assert (g.N, g.K, g.A, g.E) == (r.N, r.K, r.A, r.E)
    for name in ("rowptr_dst", "rowptr_src"):
        assert torch.equal(getattr(g, name), getattr(r, name)), name
    for name in ("col_dst", "col_src", "code_dst", "code_src"):
        assert torch.equal(getattr(g, name)[:g.A], getattr(r, name)[:r.A]), name
    assert torch.equal(g.tile_ptr, r.tile_ptr)
    n = int(r.tile_ptr[-1])
    assert torch.equal(g.tile_pack[:n], r.tile_pack[:n])
    for k in range(1, g.K):
        gp, gk = g.tile_list(k)
        rp, rk = r.tile_list(k)
        assert torch.equal(gp, rp), f"prefix {k} pointers"
        m = int(rp[-1])
        assert torch.equal(gk[:m], rk[:m]), f"prefix {k} entries"
        assert g.active_pairs(k) == r.active_pairs(k)
    assert torch.equal(got.batch, ref.batch)
    assert torch.equal(got.x.long().view(-1), ref.x.long().view(-1))
    assert torch.equal(got.y, ref.y)
    from kp_gnn_amd.ops import graph_ptr_of
    assert torch.equal(graph_ptr_of(got.batch, got.num_graphs), graph_ptr_of(ref.batch, ref.num_graphs))
    if got.pe_attr is not None:
        assert got.pe_attr.shape == ref.pe_attr.shape and not bool(got.pe_attr.any())
    # dictionary ids: the tuple behind every (node, hop) equals the dense peripheral attributes of the plain batch
    pd = got.peripheral_dict
    N, K = ref.peripheral_edge_attr.shape[:2]
    dense = torch.cat([ref.peripheral_edge_attr.reshape(N * K, -1), ref.peripheral_configuration_attr.reshape(N * K, -1)], dim=1)
    assert torch.equal(pd.pdict.rows[pd.uid.reshape(-1).long()].long(), dense)
    return got, ref


@pytest.mark.gpu
def test_collate_is_bit_identical_to_the_per_batch_build():
    from kp_gnn_amd.dataset import KHopDataset
    dev = torch.device("cuda:0")
    raw = molecules(300)
    ds = KHopDataset.from_collated(raw.collated(ZINC_ARGS), raw.node_ptr, dev, chunk_graphs=128)
    assert ds.G == 300 and ds.pdict is not None
    rng = np.random.default_rng(0)
    for B in (1, 2, 37, 300):
        ids = rng.permutation(300)[:B]
        _check_equal_to_plain_build(ds, raw, ids.tolist(), ZINC_ARGS)
    _check_equal_to_plain_build(ds, raw, [5, 5, 17, 5], ZINC_ARGS)          # repeats are legal (sampling with replacement)


@pytest.mark.gpu
def test_collate_odd_graphs_and_long_runs():
    """Runs longer than an entry's 6-bit multiplicity, tiles across several tiny graphs, the graph-diffusion kernel."""
    from kp_gnn_amd.dataset import KHopDataset
    dev = torch.device("cuda:0")
    raw = odd_graphs()
    for args in ((3, 10, 2, 1, 5, 5, "spd"), (4, 10, 2, 1, 5, 5, "gd")):
        ds = KHopDataset.from_collated(raw.collated(args), raw.node_ptr, dev, chunk_graphs=3)
        for ids in ([0], [1, 4, 3, 2], [6, 5, 4, 3, 2, 1, 0], [0, 5, 0]):
            got, _ = _check_equal_to_plain_build(ds, raw, ids, args)
        mult = ((got.csr.tile_pack[:int(got.csr.tile_ptr[-1])] >> 6) & 63) + 1
        assert int(mult.max()) == 64                                        # the star's runs were cut


@pytest.mark.gpu
def test_dataset_save_load_round_trip(tmp_path):
    from kp_gnn_amd.dataset import KHopDataset
    dev = torch.device("cuda:0")
    raw = molecules(40, seed0=3)
    ds = KHopDataset.from_collated(raw.collated(ZINC_ARGS), raw.node_ptr, dev)
    path = str(tmp_path / "zinc40.pt")
    ds.save(path)
    ds2 = KHopDataset.load(path, dev)
    ids = [3, 1, 39, 20]
    a, b = ds.collate(ids), ds2.collate(ids)
    for name in ("rowptr_dst", "rowptr_src", "col_dst", "col_src", "code_dst", "code_src", "tile_ptr", "tile_pack"):
        assert torch.equal(getattr(a.csr, name), getattr(b.csr, name)), name
    assert torch.equal(a.x, b.x) and torch.equal(a.y, b.y) and torch.equal(a.peripheral_dict.uid, b.peripheral_dict.uid)


@pytest.mark.gpu
def test_collated_batch_trains_like_the_plain_batch():
    """One KP-GIN+ body: forward + backward on a batch collated from the resident dataset equals the same on the plain batch
    (same kernels, same CSR; the peripheral dictionary has the dataset's rows instead of the batch's own, so the dictionary
    gradient is summed in another order: fp32 tolerance), and no step of it needs a host synchronisation."""
    from kp_gnn_amd import body as B
    from kp_gnn_amd.dataset import KHopDataset
    from kp_gnn_amd.layers import make_gnn_layer
    dev = torch.device("cuda:0")
    raw = molecules(64, seed0=5)
    K, L, H = 4, 4, 32
    args = (K, 50, 6, 3, 50, 50, "spd")
    ds = KHopDataset.from_collated(raw.collated(args), raw.node_ptr, dev)
    ns = argparse.Namespace(model_name="KPGINPlus", hidden_size=H, K=K, num_layer=L, num_hop1_edge=3, max_pe_num=50,
                            combine="geometric", eps=0., train_eps=False, aggr="add")
    torch.manual_seed(0)
    gnn = B.GNNPlus(num_layer=L, gnn_layer=make_gnn_layer(ns), JK="concat", norm_type="Batch", init_emb=B.EmbeddingEncoder(21, H),
                    residual=True, virtual_node=False, use_rd=False, num_hop1_edge=3, max_edge_count=50, max_hop_num=6,
                    max_distance_count=50, drop_prob=0.0)
    model = B.GraphRegression(gnn, "sum").to(dev).train()
    ids = [9, 3, 60, 21, 22, 23, 0, 63, 11, 40, 41, 5]
    plain = raw.subset(ids).collated(args).to(dev)
    plain.build_csr()
    res = []
    for batch in (plain, ds.collate(ids)):
        model.zero_grad(set_to_none=True)
        out = model(batch)
        loss = (out - batch.y).abs().mean()
        loss.backward()
        res.append((out.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    (o0, g0), (o1, g1) = res
    assert torch.allclose(o0, o1, rtol=1e-5, atol=1e-5)
    assert g0.keys() == g1.keys()
    gscale = max(float(v.abs().max()) for v in g0.values())
    for n in g0:
        tol = 2e-4 * max(float(g0[n].abs().max()), 0.05 * gscale) + 1e-7
        assert float((g0[n] - g1[n]).abs().max()) <= tol, n


def _plus_model(K, L, H, dev):
    from kp_gnn_amd import body as B
    from kp_gnn_amd.layers import make_gnn_layer
    ns = argparse.Namespace(model_name="KPGINPlus", hidden_size=H, K=K, num_layer=L, num_hop1_edge=3, max_pe_num=50,
                            combine="geometric", eps=0., train_eps=False, aggr="add")
    torch.manual_seed(0)
    gnn = B.GNNPlus(num_layer=L, gnn_layer=make_gnn_layer(ns), JK="concat", norm_type="Batch", init_emb=B.EmbeddingEncoder(21, H),
                    residual=True, virtual_node=False, use_rd=False, num_hop1_edge=3, max_edge_count=50, max_hop_num=6,
                    max_distance_count=50, drop_prob=0.0)
    return B.GraphRegression(gnn, "sum").to(dev).train()


def _step(model, batch):
    from kp_gnn_amd.ops_dense import regression_loss_and_grad
    score = model(batch)
    loss, dscore = regression_loss_and_grad(score, batch.y, "l1")
    params = [p for p in model.parameters() if p.requires_grad]
    grads = torch.autograd.grad(score, params, grad_outputs=dscore, allow_unused=True)
    return loss, score, [g if g is None else g.clone() for g in grads]


@pytest.mark.gpu
@pytest.mark.parametrize("Bsz", [96, 256])
def test_static_batch_dynamic_rows_equal_exact_shapes(Bsz):
    """A StaticBatch (capacity-shaped tensors, live node count on the device, every launch bounded by n_dyn) gives the same
    loss and gradients as the exact-shape batch of the same graphs - eagerly and from ONE captured hipGraph replayed on
    different batches (different N, A, entry counts).  BatchNorm statistics, weight gradients, table gradients and the
    readout all sum over rows: a single phantom row in any of them would show here."""
    from kp_gnn_amd.dataset import KHopDataset
    dev = torch.device("cuda:0")
    raw = molecules(400, seed0=21)
    K, L, H = 4, 4, 32
    args = (K, 50, 6, 3, 50, 50, "spd")
    ds = KHopDataset.from_collated(raw.collated(args), raw.node_ptr, dev)
    model = _plus_model(K, L, H, dev)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sb = ds.static_batch(Bsz)          # (96 graphs: ~2.2k nodes, the small-batch kernel choices; 256: ~5.9k, the large-batch ones)
    assert sb.N_cap > int(ds.h_nodes.mean() * Bsz)
    rng = np.random.default_rng(3)
    id_sets = [rng.permutation(400)[:Bsz] for _ in range(3)]

    def reset():
        model.load_state_dict(sd)          # (running statistics and num_batches_tracked included)

    def check(got, ref, what):
        (l0, s0, g0), (l1, s1, g1) = got, ref
        assert torch.allclose(s0, s1, rtol=2e-5, atol=2e-5), what
        assert abs(float(l0) - float(l1)) <= 2e-5 * max(1.0, abs(float(l1))), what
        gscale = max(float(g.abs().max()) for g in g1 if g is not None)
        for a, b in zip(g0, g1):
            assert (a is None) == (b is None), what
            if a is not None:
                tol = 1e-4 * max(float(b.abs().max()), 0.05 * gscale) + 1e-7
                assert float((a - b).abs().max()) <= tol, (what, float((a - b).abs().max()), tol)

    refs = []
    for ids in id_sets:
        reset()
        l, s_, g = _step(model, ds.collate(ids))
        refs.append((l.clone(), s_.detach().clone(), g))
    del l, s_, g          # (a live autograd graph from another stream makes the engine synchronise streams inside a later capture)
    # eager, dynamic rows
    with sb.dynamic():
        for ids, ref in zip(id_sets, refs):
            reset()
            sb.stage(ids)
            sb.launch_collate()
            l, s_, g = _step(model, sb.batch)
            assert sb.live[0] < sb.N_cap
            check((l, s_.detach(), g), ref, "eager static batch")
        del l, s_, g
        # one captured graph, replayed on every batch
        reset()
        sb.stage(id_sets[0])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            sb.launch_collate()
            _step(model, sb.batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            sb.launch_collate()
            out = _step(model, sb.batch)
    for ids, ref in zip(id_sets, refs):
        reset()
        sb.stage(ids)
        graph.replay()
        torch.cuda.synchronize()
        check((out[0], out[1].detach(), out[2]), ref, "replayed static batch")
    # a batch that does not fit is refused on the host, before anything is launched
    from kp_gnn_amd.dataset import CapacityError
    big = np.argsort(-ds.h_nodes)[:Bsz]
    if int(ds.h_nodes[big].sum()) > sb.N_cap:
        with pytest.raises(CapacityError):
            sb.stage(big)
