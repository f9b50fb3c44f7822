"""CPU: the oracle restatements (oracle/) against the golden vectors produced by the reference's own
files (tests/golden/make_golden.py).  Integer pre-transform: bit-exact.  Layers: fp32, rtol 1e-5."""
import os

import numpy as np
import pytest
import torch

from oracle import khop_oracle, kp_layers_oracle as LO

RTOL, ATOL = 1e-5, 2e-6


def _close(a, b, name, rtol=RTOL, atol=ATOL):
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(1.0, float(b.abs().max()))
    assert torch.allclose(a, b, rtol=rtol, atol=atol * scale), (name, float((a - b).abs().max()), scale)


# ----------------------------------------------------------------------------- pre-transform (integer, bit-exact)
def _khop_cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "khop_preprocess.npz"))
    return z, sorted({k.split("/in/")[0] for k in z.files if "/in/" in k})


def test_khop_oracle_bit_exact(golden_dir):
    z, cases = _khop_cases(golden_dir)
    assert len(cases) >= 30
    for c in cases:
        a = z["args/" + c.split("/")[0]]
        args = [int(v) for v in a[:6]] + [str(a[6])]
        ea = z[c + "/in/edge_attr"] if c + "/in/edge_attr" in z.files else None
        out = khop_oracle.extract_multi_hop_neighbors(int(z[c + "/in/num_nodes"]), z[c + "/in/edge_index"], ea, *args)
        want = sorted(k.split("/out/")[1] for k in z.files if k.startswith(c + "/out/"))
        assert want == sorted(out.keys()), c
        for k in want:
            g = z[c + "/out/" + k]
            assert g.shape == out[k].shape and np.array_equal(g, out[k]), (c, k)


def test_khop_quirks(golden_dir):
    """Q1 (pe_attr == 0), Q3 (spd: one active hop per edge; code ranges), Q8 (no-edge graph)."""
    z, cases = _khop_cases(golden_dir)
    for c in cases:
        if c + "/out/pe_attr" in z.files:
            assert not z[c + "/out/pe_attr"].any()
        if c.split("/")[0].endswith("spd") and c + "/out/edge_attr" in z.files:
            ea = z[c + "/out/edge_attr"]
            assert ((ea != 0).sum(1) == 1).all()
    assert "zinc_k8_spd/no_edges/out/peripheral_configuration" in z.files
    assert z["zinc_k8_spd/no_edges/out/peripheral_configuration"].shape == (4, 8, 6)


# ----------------------------------------------------------------------------- layers (fp32)
def _leafify(sd, trainable):
    """state_dict -> dict of leaves; only real parameters (keys of param_grads) require grad."""
    return {k: (v.clone().requires_grad_(True) if k in trainable else v.clone()) for k, v in sd.items()}


def _run_layer(case):
    kind, ctor = case["kind"], case["ctor"]
    p = _leafify(case["state_dict"], case["param_grads"])
    x = case["x"].clone().requires_grad_(True)
    periph = case.get("peripheral_attr")
    if periph is not None:
        periph = periph.clone().requires_grad_(True)
    pe = case.get("pe_attr")
    K = ctor.get("K", 1)
    if kind == "KPGIN":
        out = LO.kpgin_forward(p, x, case["edge_index"], case["edge_attr"], pe, periph, K=K,
                               combine_kind=ctor.get("combine", "geometric"))
    elif kind == "KPGINPlus":
        out = LO.kpginplus_forward(p, x, case["edge_index"], case["edge_attr"], pe, periph, K=K,
                                   combine_kind=ctor.get("combine"), training=True)
    elif kind == "KPGCN":
        out = LO.kpgcn_forward(p, x, case["edge_index"], case["edge_attr"], pe, periph, K=K,
                               combine_kind=ctor.get("combine", "geometric"))
    elif kind == "KPGraphSAGE":
        out = LO.kpgraphsage_forward(p, x, case["edge_index"], case["edge_attr"], pe, periph, K=K,
                                     combine_kind=ctor.get("combine", "geometric"))
    elif kind == "GINE":
        out = LO.gine_forward(p, x, case["edge_index"], case["edge_attr"], training=True)
    (out * case["out_weight"]).sum().backward()
    return p, x, periph, out


def test_layer_oracle_matches_reference(golden_dir):
    cases = torch.load(os.path.join(golden_dir, "layers.pt"), weights_only=True)
    assert len(cases) >= 22
    for name, case in cases.items():
        p, x, periph, out = _run_layer(case)
        _close(out.detach(), case["out"], name + ":out")
        _close(x.grad, case["grad_x"], name + ":grad_x", rtol=1e-4)
        if periph is not None:
            _close(periph.grad, case["grad_peripheral_attr"], name + ":grad_periph", rtol=1e-4)
        # parameter grads: absolute tolerance relative to the largest grad of the case (a Linear bias in
        # front of BatchNorm has an analytically-zero grad that is pure rounding noise in both runs)
        gscale = max(float(g.abs().max()) for g in case["param_grads"].values())
        for k, g in case["param_grads"].items():
            got = p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])
            assert torch.allclose(got, g, rtol=1e-4, atol=1e-5 * max(1.0, gscale)), \
                (name, k, float((got - g).abs().max()), gscale)
        for k, v in case["state_dict_after"].items():
            if "running" in k:
                _close(p[k], v, f"{name}:{k}")


def test_combine_oracle_matches_reference(golden_dir):
    cases = torch.load(os.path.join(golden_dir, "combine.pt"), weights_only=True)
    for name, case in cases.items():
        p = _leafify(case["state_dict"], case["param_grads"])
        x = case["x"].clone().requires_grad_(True)
        out = LO.attention_combine(p, x) if name.startswith("att") else LO.geometric_combine(p, x)
        (out * case["out_weight"]).sum().backward()
        _close(out.detach(), case["out"], name + ":out")
        _close(x.grad, case["grad_x"], name + ":grad_x", rtol=1e-4)
        for k, g in case["param_grads"].items():
            _close(p[k].grad, g, f"{name}:grad[{k}]", rtol=1e-4, atol=1e-5)


def test_path_encoding_table_never_trains(golden_dir):
    """Q1: pe_attr is all-zero, row 0 is the padding row -> hopk_node_path_emb gets exactly zero grad."""
    cases = torch.load(os.path.join(golden_dir, "layers.pt"), weights_only=True)
    for name, case in cases.items():
        g = case["param_grads"].get("hopk_node_path_emb.weight")
        if g is not None:
            assert not g.any(), name


# ----------------------------------------------------------------------------- whole bodies (fp32)
BODY_KIND = {"KPGINPlus": ("GNNPlus", "KPGINPlus"), "KPGIN": ("GNN", "KPGIN"), "KPGCN": ("GNN", "KPGCN"),
             "KPGINPrime": ("GNNPrime", "KPGIN")}


def test_body_oracle_matches_reference(golden_dir):
    from oracle import kp_model_oracle as MO
    cases = torch.load(os.path.join(golden_dir, "bodies.pt"), weights_only=True)
    assert len(cases) >= 6
    for name, case in cases.items():
        kind, layer_kind = BODY_KIND[case["model_name"]]
        p = _leafify(case["state_dict"], case["param_grads"])
        score = MO.graph_regression_forward(p, case["inputs"], kind=kind, layer_kind=layer_kind, K=case["K"],
                                            num_layer=case["L"], combine_kind=case["combine"], JK=case["JK"],
                                            residual=bool(case["residual"]), virtual_node=bool(case["virtual_node"]),
                                            training=True)
        loss = (score.squeeze() - case["y"].squeeze()).abs().mean()
        loss.backward()
        _close(score.detach(), case["score"], name + ":score", rtol=1e-4, atol=1e-5)
        _close(loss.detach(), case["loss"], name + ":loss", rtol=1e-4, atol=1e-5)
        gscale = max(float(g.abs().max()) for g in case["param_grads"].values())
        for k, g in case["param_grads"].items():
            got = p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])
            assert torch.allclose(got, g, rtol=1e-3, atol=2e-5 * max(1.0, gscale)), \
                (name, k, float((got - g).abs().max()), gscale)


def test_kgin_oracle_matches_reference_goldens(golden_dir):
    """oracle.kgin_forward (run_simulation.py's mask-only KGINConv) against vectors produced by the reference's own class
    (cut out of run_simulation.py with ast and executed alone, tests/golden/make_golden.py): output and every gradient."""
    cases = torch.load(os.path.join(golden_dir, "kgin.pt"), weights_only=True)
    assert len(cases) >= 3
    for name, c in cases.items():
        p = {k: v.clone().requires_grad_(v.is_floating_point() and k != "eps") for k, v in c["state_dict"].items()}
        x = c["x"].clone().requires_grad_(True)
        out = LO.kgin_forward(p, x, c["edge_index"], c["edge_attr"], K=c["K"], batch=c["batch"] if c["pool"] else None)
        (out * c["out_weight"]).sum().backward()
        _close(out, c["out"], name + ":out")
        _close(x.grad, c["grad_x"], name + ":grad_x")
        for k, g in c["param_grads"].items():
            _close(p[k].grad, g, f"{name}:grad[{k}]")
