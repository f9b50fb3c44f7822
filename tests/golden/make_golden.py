#!/usr/bin/env python3
"""Generate the golden parity vectors under tests/golden/ by running the REFERENCE's own files.

Run in the build container only (needs /root/reference, read-only):

    python tests/golden/make_golden.py

How it works: the reference's layer / model / preprocessing files are imported unmodified from
/root/reference.  The only thing they lack here is torch_geometric (third party, PyG 2.1.0 pin,
not installed, not fetchable), so tests/pyg_standin/ supplies a test-only restatement of the few
PyG symbols they touch (documented contract of MessagePassing.propagate etc., see that package's
docstring).  The vectors therefore pin every line of reference-owned arithmetic; PyG's own
gather / scatter-sum is pinned only through its documented contract.

Only *data* is written: inputs, seeded weights, outputs and gradients as tensors.  No reference
source text or bytecode is stored (sys.dont_write_bytecode is set and nothing is copied).
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("KPGNN_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "..", "pyg_standin"))
sys.path.insert(0, REF)

import argparse  # noqa: E402

import networkx as nx  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch_geometric.data import Batch, Data  # noqa: E402  (stand-in)

import data_utils as ref_data_utils  # noqa: E402  (reference)
from layers.combine import AttentionCombine, GeometricCombine  # noqa: E402
from layers.gine import GINEConv  # noqa: E402
from layers.input_encoder import EmbeddingEncoder  # noqa: E402
from layers.KPGCN import KPGCNConv  # noqa: E402
from layers.KPGIN import KPGINConv  # noqa: E402
from layers.KPGINplus import KPGINPlusConv  # noqa: E402
from layers.KPGraphSAGE import KPGraphSAGEConv  # noqa: E402
from layers.layer_utils import make_gnn_layer  # noqa: E402
from models.GraphRegression import GraphRegression  # noqa: E402
from models.model_utils import make_GNN  # noqa: E402

torch.set_num_threads(1)


# ----------------------------------------------------------------------------- input graphs
def molecule_like(seed):
    """ZINC-shaped random molecule graph."""
    rng = np.random.default_rng(seed)
    n = int(np.clip(round(rng.normal(23.2, 4.5)), 9, 37))
    G = nx.Graph()
    G.add_node(0)
    for v in range(1, n):
        cand = [u for u in G.nodes if G.degree(u) < 3]
        w = np.array([1 + 3 * (u / max(1, v - 1)) for u in cand], dtype=np.float64)
        u = int(rng.choice(cand, p=w / w.sum()))
        G.add_edge(u, v)
    target = int(rng.integers(1, 4))
    pairs = []
    for a, dd in nx.all_pairs_shortest_path_length(G, cutoff=5):
        for b, dist in dd.items():
            if a < b and dist in (4, 5):
                pairs.append((a, b))
    rng.shuffle(pairs)
    added = 0
    for a, b in pairs:
        if added >= target:
            break
        if G.degree(a) < 3 and G.degree(b) < 3 and not G.has_edge(a, b):
            G.add_edge(a, b)
            added += 1
    bond = {}
    for (a, b) in G.edges:
        bond[(a, b)] = bond[(b, a)] = int(rng.choice([1, 2, 3], p=[.75, .2, .05]))
    dir_edges = list(G.to_directed().edges)
    ei = torch.tensor(dir_edges, dtype=torch.long).t().contiguous()
    ea = torch.tensor([bond[e] for e in dir_edges], dtype=torch.long)
    x = torch.from_numpy(rng.integers(0, 21, size=(n, 1))).long()
    return x, ei, ea


def nx_to_edge_index(G):
    e = list(G.to_directed().edges) if not G.is_directed() else list(G.edges)
    if len(e) == 0:
        return torch.zeros([2, 0], dtype=torch.long)
    return torch.tensor(e, dtype=torch.long).t().contiguous()


def input_graphs():
    """name -> (x, edge_index, edge_attr-or-None)."""
    out = {}
    sr = nx.read_graph6(os.path.join(REF, "data/sr25/raw/sr251256.g6"))
    for i in (0, 7):
        out[f"sr25_{i}"] = (torch.ones(25, 1), nx_to_edge_index(sr[i]), None)
    for n, s in ((20, 0), (40, 1), (64, 2)):
        G = nx.random_regular_graph(d=3, n=n, seed=s)
        out[f"reg3_n{n}_s{s}"] = (torch.ones(n, 1), nx_to_edge_index(G), None)
    for s in range(6):
        x, ei, ea = molecule_like(s)
        out[f"mol_{s}"] = (x, ei, ea + 1)  # edge_feature_transform, train_ZINC.py:96-99
    # edge cases
    out["path5"] = (torch.ones(5, 1), nx_to_edge_index(nx.path_graph(5)), None)
    out["two_components"] = (torch.ones(9, 1), nx_to_edge_index(
        nx.disjoint_union(nx.cycle_graph(5), nx.star_graph(3))), None)
    G = nx.cycle_graph(6)
    G.add_node(6)
    G.add_node(7)  # isolated nodes at the end
    out["isolated_tail"] = (torch.ones(8, 1), nx_to_edge_index(G), None)
    out["no_edges"] = (torch.ones(4, 1), torch.zeros([2, 0], dtype=torch.long), None)
    D = nx.DiGraph()
    D.add_edges_from([(0, 1), (1, 2), (2, 0), (2, 3), (3, 4), (4, 2), (1, 4), (5, 0)])
    out["directed6"] = (torch.ones(6, 1), nx_to_edge_index(D), torch.tensor([2, 3, 2, 4, 2, 3, 2, 4]))
    out["complete6_typed"] = (torch.ones(6, 1), nx_to_edge_index(nx.complete_graph(6)),
                              torch.tensor([2 + ((a + b) % 3) for a, b in nx.complete_graph(6).to_directed().edges]))
    return out


PRE_ARGS = {
    # name: (K, max_edge_attr_num, max_hop_num, max_edge_type, max_edge_count, max_distance_count, kernel)
    "zinc_k8_spd": (8, 50, 6, 3, 50, 50, "spd"),      # train_ZINC.py:125-134,191-194
    "zinc_k16_gd": (16, 50, 6, 3, 50, 50, "gd"),     # README.md:128
    "zinc_k3_gd": (3, 50, 6, 3, 50, 50, "gd"),
    "zinc_k1_spd": (1, 50, 6, 3, 50, 50, "spd"),
    "sr_k4_spd": (4, 1000, 4, 1, 1000, 1000, "spd"),  # train_SR.py:115-125
    "sr_k4_gd": (4, 1000, 4, 1, 1000, 1000, "gd"),
    "sim_k4_spd": (4, 10, 1, 1, 1, 1, "spd"),         # run_simulation.py:103
    "sim_k8_spd": (8, 10, 1, 1, 1, 1, "spd"),
    "qm9_k6_spd": (6, 50, 5, 4, 20, 15, "spd"),       # train_qm9.py:141-158
    "tu_k3_spd": (3, 10, 3, 1, 30, 30, "spd"),        # KP-GCN / MUTAG shaped
    "clamp_k6_gd": (6, 3, 2, 2, 2, 3, "gd"),          # tiny clamps so every clamp branch fires
}


def run_pretransform(x, ei, ea, args):
    d = Data(x=x.clone(), edge_index=ei.clone(), edge_attr=None if ea is None else ea.clone())
    d = ref_data_utils.extract_multi_hop_neighbors(d, *args)
    res = {}
    for k in ("edge_index", "edge_attr", "pe_attr", "peripheral_edge_attr", "peripheral_configuration_attr",
              "peripheral_configuration"):
        if k in d:
            res[k] = getattr(d, k)
    return res


def make_preprocess_goldens(outdir):
    graphs = input_graphs()
    plan = {
        "zinc_k8_spd": ["mol_0", "mol_1", "mol_2", "mol_3", "mol_4", "mol_5", "no_edges", "isolated_tail", "directed6"],
        "zinc_k16_gd": ["mol_0", "mol_1", "two_components"],  # (walk counts must stay < 2^31: the reference's .int() wraps beyond)
        "zinc_k3_gd": ["mol_2", "two_components", "directed6"],
        "zinc_k1_spd": ["mol_3", "path5"],
        "sr_k4_spd": ["sr25_0", "sr25_7"],
        "sr_k4_gd": ["sr25_0"],
        "sim_k4_spd": ["reg3_n20_s0", "reg3_n40_s1"],
        "sim_k8_spd": ["reg3_n64_s2", "path5"],
        "qm9_k6_spd": ["mol_4", "complete6_typed", "two_components"],
        "tu_k3_spd": ["reg3_n20_s0", "path5", "isolated_tail", "two_components"],
        "clamp_k6_gd": ["mol_5", "complete6_typed", "sr25_0", "directed6"],
    }
    blob = {}
    n = 0
    for aname, gnames in plan.items():
        args = PRE_ARGS[aname]
        for gname in gnames:
            x, ei, ea = graphs[gname]
            res = run_pretransform(x, ei, ea, args)
            key = f"{aname}/{gname}"
            blob[key + "/in/num_nodes"] = np.int64(x.size(0))
            blob[key + "/in/edge_index"] = ei.numpy().astype(np.int32)
            if ea is not None:
                blob[key + "/in/edge_attr"] = ea.numpy().astype(np.int32)
            for k, v in res.items():
                blob[key + "/out/" + k] = v.numpy().astype(np.int32)
            n += 1
    for aname, a in PRE_ARGS.items():
        blob["args/" + aname] = np.array([str(v) for v in a])
    np.savez_compressed(os.path.join(outdir, "khop_preprocess.npz"), **blob)
    print(f"khop_preprocess.npz: {n} cases")


# ----------------------------------------------------------------------------- layers
def batch_of(names, args, int_x=False):
    graphs = input_graphs()
    lst = []
    for nme in names:
        x, ei, ea = graphs[nme]
        if int_x and x.dtype != torch.long:  # EmbeddingEncoder(21, h) wants atom-type indices
            x = (torch.arange(x.size(0)) * 7 % 21).view(-1, 1)
        d = Data(x=x.clone(), edge_index=ei.clone(), edge_attr=None if ea is None else ea.clone())
        d = ref_data_utils.extract_multi_hop_neighbors(d, *args)
        lst.append(d)
    return Batch.from_data_list(lst)


def grads_of(module):
    return {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for k, p in module.named_parameters()}


def layer_case(kind, ctor_kw, batch_names, pre, seed, with_periph=True, K_used=None):
    """Run one reference layer fwd+bwd; returns a dict of tensors."""
    torch.manual_seed(seed)
    b = batch_of(batch_names, PRE_ARGS[pre])
    K = ctor_kw["K"] if "K" in ctor_kw else 1
    if kind == "KPGIN":
        layer = KPGINConv(**ctor_kw)
        D = ctor_kw["input_size"] // K
        xshape = [b.num_nodes, ctor_kw["input_size"]]
    elif kind == "KPGINPlus":
        layer = KPGINPlusConv(**ctor_kw)
        D = ctor_kw["input_size"]
        xshape = [b.num_nodes, K, D]
    elif kind == "KPGCN":
        layer = KPGCNConv(**ctor_kw)
        D = ctor_kw["output_size"] // K
        xshape = [b.num_nodes, ctor_kw["input_size"]]
    elif kind == "KPGraphSAGE":
        layer = KPGraphSAGEConv(**ctor_kw)
        D = ctor_kw["input_size"] // K
        xshape = [b.num_nodes, ctor_kw["input_size"]]
    elif kind == "GINE":
        layer = GINEConv(**ctor_kw)
        D = ctor_kw["input_size"]
        xshape = [b.num_nodes, D]
    else:
        raise ValueError(kind)
    # give every parameter a non-trivial value (alphas / eps start at 0 in reset_parameters)
    with torch.no_grad():
        for name, p in layer.named_parameters():
            if name.endswith("alphas"):
                p.copy_(torch.randn_like(p) * 0.7)
            if name == "eps":
                p.fill_(0.25)
        for name, buf in layer.named_buffers():
            if name == "eps":
                buf.fill_(ctor_kw.get("eps", 0.0))
    layer.train()
    x = torch.randn(xshape)
    x_in = x.clone().requires_grad_(True)
    edge_attr = b.edge_attr[:, :K]
    pe_attr = None if K == 1 or "pe_attr" not in b else b.pe_attr[:, :K - 1]
    periph = None
    if with_periph and kind != "GINE":
        periph = (torch.randn(b.num_nodes, K, D) * 0.5).requires_grad_(True)
    sd_before = {k: v.clone() for k, v in layer.state_dict().items()}
    # the reference mutates its input view in place (adds exact zeros, Q1) -> feed a non-leaf clone
    xin2 = x_in * 1.0
    if kind == "GINE":
        out = layer(xin2, b.edge_index, edge_attr)
    else:
        out = layer(xin2, b.edge_index, edge_attr, pe_attr, periph)
    w = torch.randn_like(out)
    loss = (out * w).sum()
    loss.backward()
    case = {
        "kind": kind, "pre": pre, "ctor": {k: (v if not isinstance(v, bool) else int(v)) for k, v in ctor_kw.items()},
        "x": x, "edge_index": b.edge_index, "edge_attr": edge_attr.contiguous(),
        "out": out.detach(), "out_weight": w, "grad_x": x_in.grad.clone(),
        "state_dict": sd_before, "param_grads": grads_of(layer),
        "state_dict_after": {k: v.clone() for k, v in layer.state_dict().items() if "running" in k or "num_batches" in k},
    }
    if pe_attr is not None:
        case["pe_attr"] = pe_attr.contiguous()
    if periph is not None:
        case["peripheral_attr"] = periph.detach().clone()
        case["grad_peripheral_attr"] = periph.grad.clone()
    return case


def make_layer_goldens(outdir):
    cases = {}
    mols2 = ["mol_0", "mol_1"]
    mols3 = ["mol_2", "mol_3", "mol_4"]
    # --- KP-GIN (layers/KPGIN.py)
    cases["kpgin_k8_h104_geo_spd"] = layer_case(
        "KPGIN", dict(input_size=104, output_size=104, K=8, eps=0., train_eps=False, num_hop1_edge=3, num_pe=50,
                      combine="geometric"), mols3, "zinc_k8_spd", 1)
    cases["kpgin_k8_h104_att_spd"] = layer_case(
        "KPGIN", dict(input_size=104, output_size=104, K=8, eps=0., train_eps=True, num_hop1_edge=3, num_pe=50,
                      combine="attention"), mols2, "zinc_k8_spd", 2)
    cases["kpgin_k16_h96_geo_gd"] = layer_case(
        "KPGIN", dict(input_size=96, output_size=96, K=16, eps=0.1, train_eps=False, num_hop1_edge=3, num_pe=50,
                      combine="geometric"), mols2, "zinc_k16_gd", 3)
    cases["kpgin_k4_h8_att_gd_sr"] = layer_case(
        "KPGIN", dict(input_size=8, output_size=8, K=4, eps=0., train_eps=False, num_hop1_edge=1, num_pe=1000,
                      combine="attention"), ["sr25_0"], "sr_k4_gd", 4)
    cases["kpgin_k1_h16"] = layer_case(
        "KPGIN", dict(input_size=16, output_size=16, K=1, num_hop1_edge=3, num_pe=50), ["mol_3", "path5"],
        "zinc_k1_spd", 5, with_periph=False)
    cases["kpgin_k6_h120_geo_qm9"] = layer_case(
        "KPGIN", dict(input_size=120, output_size=120, K=6, num_hop1_edge=4, num_pe=50, combine="geometric"),
        ["mol_4", "two_components"], "qm9_k6_spd", 6)
    cases["kpgin_k3_in12_out24_directed"] = layer_case(
        "KPGIN", dict(input_size=12, output_size=24, K=3, num_hop1_edge=3, num_pe=50, combine="geometric"),
        ["directed6", "mol_2"], "zinc_k3_gd", 7)
    # --- KP-GIN+ (layers/KPGINplus.py)
    cases["kpginplus_k8_h104_geo_spd"] = layer_case(
        "KPGINPlus", dict(input_size=104, output_size=104, K=8, num_hop1_edge=3, num_pe=50, combine="geometric"),
        mols2, "zinc_k8_spd", 11)
    cases["kpginplus_k8_h40_att_spd"] = layer_case(
        "KPGINPlus", dict(input_size=40, output_size=40, K=8, num_hop1_edge=3, num_pe=50, combine="attention"),
        mols3, "zinc_k8_spd", 12)
    cases["kpginplus_k3_h16_geo_prefix"] = layer_case(  # GNNPlus layer 3: edge_attr[:, :3] of a K=8 list (Q12)
        "KPGINPlus", dict(input_size=16, output_size=16, K=3, num_hop1_edge=3, num_pe=50, combine="geometric"),
        mols2, "zinc_k8_spd", 13)
    cases["kpginplus_k1_h16"] = layer_case(
        "KPGINPlus", dict(input_size=16, output_size=16, K=1, num_hop1_edge=3, num_pe=50, combine="geometric"),
        mols2, "zinc_k8_spd", 14)
    cases["kpginplus_k4_h12_att_gd_sr"] = layer_case(
        "KPGINPlus", dict(input_size=12, output_size=12, K=4, num_hop1_edge=1, num_pe=1000, combine="attention"),
        ["sr25_7"], "sr_k4_spd", 15)
    cases["kpginplus_k16_h24_geo_gd"] = layer_case(
        "KPGINPlus", dict(input_size=24, output_size=24, K=16, num_hop1_edge=3, num_pe=50, combine="geometric"),
        ["mol_0", "two_components"], "zinc_k16_gd", 16)
    # --- KP-GCN (layers/KPGCN.py)
    cases["kpgcn_k3_h33_geo"] = layer_case(
        "KPGCN", dict(input_size=33, output_size=33, K=3, num_hop1_edge=1, num_pe=10, combine="geometric"),
        ["reg3_n20_s0", "path5", "isolated_tail", "two_components"], "tu_k3_spd", 21)
    cases["kpgcn_k3_h33_att"] = layer_case(
        "KPGCN", dict(input_size=33, output_size=33, K=3, num_hop1_edge=1, num_pe=10, combine="attention"),
        ["reg3_n20_s0", "two_components"], "tu_k3_spd", 22)
    cases["kpgcn_k1_h8"] = layer_case(
        "KPGCN", dict(input_size=8, output_size=8, K=1, num_hop1_edge=1, num_pe=10), ["path5", "reg3_n20_s0"],
        "tu_k3_spd", 23, with_periph=False)
    cases["kpgcn_k8_in20_out104_zinc"] = layer_case(
        "KPGCN", dict(input_size=20, output_size=104, K=8, num_hop1_edge=3, num_pe=50, combine="geometric"),
        mols2, "zinc_k8_spd", 24)
    # --- GINE over the K-hop edge list masked by column 0 (layers/gine.py:49-59, GNNs.py:679)
    cases["gine_h96_on_k16_gd"] = layer_case(
        "GINE", dict(input_size=96, output_size=96, eps=0., num_hop1_edge=3, train_eps=False), mols2, "zinc_k16_gd", 31)
    cases["gine_h16_train_eps"] = layer_case(
        "GINE", dict(input_size=16, output_size=24, eps=0.2, num_hop1_edge=3, train_eps=True), mols3, "zinc_k8_spd", 32)
    # --- KP-GraphSAGE (layers/KPGraphSAGE.py)
    cases["kpsage_k3_h33_geo"] = layer_case(
        "KPGraphSAGE", dict(input_size=33, output_size=33, K=3, aggr="add", num_hop1_edge=1, num_pe=10,
                            combine="geometric"), ["reg3_n20_s0", "two_components"], "tu_k3_spd", 41)
    cases["kpsage_k8_in104_out40_att"] = layer_case(
        "KPGraphSAGE", dict(input_size=104, output_size=40, K=8, aggr="add", num_hop1_edge=3, num_pe=50,
                            combine="attention"), mols2, "zinc_k8_spd", 42)
    cases["kpsage_k1_h8"] = layer_case(
        "KPGraphSAGE", dict(input_size=8, output_size=8, K=1, aggr="add", num_hop1_edge=1, num_pe=10),
        ["path5", "reg3_n20_s0"], "tu_k3_spd", 43, with_periph=False)
    torch.save(cases, os.path.join(outdir, "layers.pt"))
    print(f"layers.pt: {len(cases)} cases")


# ----------------------------------------------------------------------------- combine modules alone
def make_combine_goldens(outdir):
    cases = {}
    for name, (N, K, D, seed) in {"att_n37_k8_d104": (37, 8, 104, 41), "att_n50_k16_d6": (50, 16, 6, 42),
                                  "att_n9_k3_d11": (9, 3, 11, 43), "att_n5_k2_d1": (5, 2, 1, 44)}.items():
        torch.manual_seed(seed)
        m = AttentionCombine(D, K)
        x = torch.randn(N, K, D, requires_grad=True)
        out = m(x)
        w = torch.randn_like(out)
        (out * w).sum().backward()
        cases[name] = {"x": x.detach().clone(), "out": out.detach(), "out_weight": w, "grad_x": x.grad.clone(),
                       "state_dict": {k: v.clone() for k, v in m.state_dict().items()}, "param_grads": grads_of(m)}
    for name, (N, K, D, seed) in {"geo_n37_k8_d104": (37, 8, 104, 51), "geo_n50_k16_d6": (50, 16, 6, 52),
                                  "geo_n4_k2_d3": (4, 2, 3, 53)}.items():
        torch.manual_seed(seed)
        m = GeometricCombine(K, D)
        with torch.no_grad():
            m.alphas.copy_(torch.randn(D))
        x = torch.randn(N, K, D, requires_grad=True)
        out = m(x)
        w = torch.randn_like(out)
        (out * w).sum().backward()
        cases[name] = {"x": x.detach().clone(), "out": out.detach(), "out_weight": w, "grad_x": x.grad.clone(),
                       "state_dict": {k: v.clone() for k, v in m.state_dict().items()}, "param_grads": grads_of(m)}
    torch.save(cases, os.path.join(outdir, "combine.pt"))
    print(f"combine.pt: {len(cases)} cases")


# ----------------------------------------------------------------------------- whole bodies
def body_case(model_name, K, L, h, combine, pre, names, seed, residual=True, JK="concat", virtual_node=False,
              num_hop1_edge=3, max_pe_num=50, max_edge_count=50, max_hop_num=6, max_distance_count=50):
    torch.manual_seed(seed)
    args = argparse.Namespace(model_name=model_name, hidden_size=h, K=K, num_layer=L, num_hop1_edge=num_hop1_edge,
                              max_pe_num=max_pe_num, combine=combine, eps=0., train_eps=False, aggr="add")
    layer = make_gnn_layer(args)
    init_emb = EmbeddingEncoder(21, h)
    GNNModel = make_GNN(args)
    gnn = GNNModel(num_layer=L, gnn_layer=layer, JK=JK, norm_type="Batch", init_emb=init_emb, residual=residual,
                   virtual_node=virtual_node, use_rd=False, num_hop1_edge=num_hop1_edge, max_edge_count=max_edge_count,
                   max_hop_num=max_hop_num, max_distance_count=max_distance_count, wo_peripheral_edge=False,
                   wo_peripheral_configuration=False, drop_prob=0.0)
    model = GraphRegression(embedding_model=gnn, pooling_method="sum")
    model.reset_parameters()
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("alphas"):
                p.copy_(torch.randn_like(p) * 0.5)
    model.train()
    b = batch_of(names, PRE_ARGS[pre], int_x=True)
    y = torch.randn(len(names))
    sd_before = {k: v.clone() for k, v in model.state_dict().items()}
    inputs = {k: getattr(b, k).clone() for k in ("x", "edge_index", "edge_attr", "pe_attr", "peripheral_edge_attr",
                                                 "peripheral_configuration_attr", "batch")}
    score = model(b)
    loss = (score.squeeze() - y.squeeze()).abs().mean()  # train_ZINC.py:42
    loss.backward()
    return {"model_name": model_name, "K": K, "L": L, "h": h, "combine": combine, "residual": int(residual), "JK": JK,
            "virtual_node": int(virtual_node),
            "hparams": dict(num_hop1_edge=num_hop1_edge, max_pe_num=max_pe_num, max_edge_count=max_edge_count,
                            max_hop_num=max_hop_num, max_distance_count=max_distance_count),
            "inputs": inputs, "y": y, "score": score.detach(), "loss": loss.detach(), "state_dict": sd_before,
            "param_grads": grads_of(model),
            "state_dict_after": {k: v.clone() for k, v in model.state_dict().items() if "running" in k}}


def make_body_goldens(outdir):
    cases = {}
    cases["gnnplus_k8_l8_h104_geo"] = body_case("KPGINPlus", 8, 8, 104, "geometric", "zinc_k8_spd",
                                                ["mol_0", "mol_1", "mol_2"], 61)
    cases["gnnplus_k3_l4_h16_att"] = body_case("KPGINPlus", 3, 4, 16, "attention", "zinc_k3_gd",
                                               ["mol_2", "mol_3", "directed6", "two_components"], 62)
    cases["gnn_kpgin_k8_l3_h104_geo"] = body_case("KPGIN", 8, 3, 104, "geometric", "zinc_k8_spd",
                                                  ["mol_0", "mol_4"], 63)
    cases["gnn_kpgin_k4_l2_h16_att_vn"] = body_case("KPGIN", 3, 2, 18, "attention", "zinc_k3_gd",
                                                    ["mol_1", "mol_5", "directed6"], 64, virtual_node=True, JK="last")
    cases["gnn_kpgcn_k3_l2_h33_geo"] = body_case("KPGCN", 3, 2, 33, "geometric", "zinc_k3_gd",
                                                 ["mol_2", "mol_3"], 65, residual=False, JK="sum")
    cases["gnnprime_k3_l3_h18_geo"] = body_case("KPGINPrime", 3, 3, 18, "geometric", "zinc_k3_gd",
                                                ["mol_0", "mol_3"], 66)
    torch.save(cases, os.path.join(outdir, "bodies.pt"))
    print(f"bodies.pt: {len(cases)} cases")


# ----------------------------------------------------------------------------- run_simulation.py's KGINConv
def reference_kgin_class(graph_pool):
    """The reference's `KGINConv` (run_simulation.py:29-93) WITHOUT running the script: run_simulation.py has no
    `__main__` guard around its experiment, so importing it would run the whole simulation.  The class definition is cut
    out of the parsed source (ast) and executed alone, in a namespace that holds what the script's imports would have
    given it (torch, nn, F, math, the stand-in MessagePassing / global_add_pool) and the module-global `args.graph` the
    class reads (:83).  Nothing of the source text is stored."""
    import ast
    import math
    import torch.nn as nn
    import torch.nn.functional as F
    from torch_geometric.nn import MessagePassing, global_add_pool
    path = os.path.join(REF, "run_simulation.py")
    tree = ast.parse(open(path).read(), filename=path)
    node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "KGINConv"]
    assert len(node) == 1
    ns = {"torch": torch, "nn": nn, "F": F, "math": math, "MessagePassing": MessagePassing,
          "global_add_pool": global_add_pool, "args": argparse.Namespace(graph=graph_pool)}
    exec(compile(ast.Module(body=node, type_ignores=[]), path, "exec"), ns)
    return ns["KGINConv"]


def make_kgin_goldens(outdir):
    graphs = input_graphs()
    cases = {}
    for name, gnames, pre, K, hs, pool, seed in (("kgin_reg40_k4", ["reg3_n40_s1"], "sim_k4_spd", 4, 16, False, 1),
                                                 ("kgin_reg64_k8", ["reg3_n64_s2"], "sim_k8_spd", 8, 16, False, 2),
                                                 ("kgin_2graphs_k4_pool", ["reg3_n20_s0", "reg3_n40_s1"], "sim_k4_spd", 4, 8, True, 3)):
        datas = []
        for gn in gnames:
            x, ei, ea = graphs[gn]
            r = run_pretransform(x, ei, ea, PRE_ARGS[pre])
            datas.append(Data(x=x.clone(), edge_index=r["edge_index"], edge_attr=r["edge_attr"]))
        b = Batch.from_data_list(datas)
        torch.manual_seed(seed)
        layer = reference_kgin_class(pool)(hs, K)
        sd = {k: v.detach().clone() for k, v in layer.state_dict().items()}
        gen = torch.Generator().manual_seed(seed)
        x = (b.x + 0.1 * torch.randn(b.x.shape, generator=gen)).requires_grad_(True)
        out = layer(x, b.edge_index, b.edge_attr, b.batch)
        w = torch.randn(out.shape, generator=gen)
        (out * w).sum().backward()
        cases[name] = {"K": K, "hidden_size": hs, "pool": pool, "state_dict": sd, "x": x.detach().clone(),
                       "edge_index": b.edge_index.clone(), "edge_attr": b.edge_attr.clone(), "batch": b.batch.clone(),
                       "out": out.detach().clone(), "out_weight": w, "grad_x": x.grad.clone(), "param_grads": grads_of(layer)}
    torch.save(cases, os.path.join(outdir, "kgin.pt"))
    print(f"kgin.pt: {len(cases)} cases")


if __name__ == "__main__":
    which = sys.argv[1:] or ["pre", "layers", "combine", "bodies", "kgin"]
    if "kgin" in which:
        make_kgin_goldens(HERE)
    if "pre" in which:
        make_preprocess_goldens(HERE)
    if "combine" in which:
        make_combine_goldens(HERE)
    if "layers" in which:
        make_layer_goldens(HERE)
    if "bodies" in which:
        make_body_goldens(HERE)
