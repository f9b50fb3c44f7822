"""ORACLE (test infrastructure, not product code): plain-PyTorch CPU restatement of the reference's
model bodies + graph-regression head, on top of oracle/kp_layers_oracle.py.  This is the whole-model CPU
path that bench.py times as `cpu_baseline` ("port") and that smoke() checks the GPU result against.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Functional: `p` is a state_dict-like dict with the reference's key names
(`embedding_model.gnns.0.mlp.0.weight`, ...).  Follows
    /root/reference/models/GNNs.py:142-235   GNN.forward        -> body_forward(kind="GNN")
    /root/reference/models/GNNs.py:363-474   GNNPlus.forward    -> body_forward(kind="GNNPlus")
    /root/reference/models/GNNs.py:606-722   GNNPrime.forward   -> body_forward(kind="GNNPrime")
    /root/reference/models/GraphRegression.py:46-51             -> graph_regression_forward
    /root/reference/layers/feature_encoder.py:62-67, input_encoder.py:21-22
Parity pin: tests/golden/bodies.pt (reference GNN / GNNPlus / GNNPrime + GraphRegression run by
tests/golden/make_golden.py); tests/test_oracle_golden.py compares score, loss and every parameter grad.
"""
import torch
import torch.nn.functional as F

from . import kp_layers_oracle as LO

sub = LO.sub


def feature_concat_encoder(p, x):
    n = x.shape[-1]
    # NB the bodies construct this encoder with `padding=0` (GNNs.py:91,96), which is falsy: NO padding row
    cols = [F.embedding(x[..., i], p[f"embedding_list.{i}.weight"]) for i in range(n)]
    return F.linear(torch.cat(cols, dim=-1), p["proj.weight"], p["proj.bias"])


def batch_norm(p, prefix, x, training):
    return F.batch_norm(x, p.get(prefix + ".running_mean"), p.get(prefix + ".running_var"), p[prefix + ".weight"],
                        p[prefix + ".bias"], training, 0.1, 1e-5)


def global_add_pool(x, batch, size):
    return x.new_zeros((size,) + tuple(x.shape[1:])).index_add_(0, batch, x)


def _vn_mlp(p, prefix, x, training):
    return LO.mlp_bn(p, prefix, x, training)


def body_forward(p, data, *, kind, layer_kind, K, num_layer, combine_kind, JK="concat", residual=True,
                 virtual_node=False, num_l1_layer=1, training=True):
    """data: dict with x, edge_index, edge_attr, batch and optionally pe_attr, peripheral_edge_attr,
    peripheral_configuration_attr.  layer_kind in {KPGIN, KPGCN, KPGINPlus}.  Dropout prob 0 assumed."""
    edge_index, edge_attr, batch = data["edge_index"], data["edge_attr"], data["batch"]
    pe_attr = data.get("pe_attr")
    if "init_proj.z_embedding.weight" in p:      # QM9InputEncoder (layers/input_encoder.py:63-84)
        xin = data["x"]
        if data.get("z") is not None:
            xin = torch.cat([F.embedding(data["z"], p["init_proj.z_embedding.weight"]), xin], -1)
        x = F.linear(xin, p["init_proj.init_proj.weight"], p["init_proj.init_proj.bias"])
    else:
        x = F.embedding(data["x"], p["init_proj.init_proj.weight"]).squeeze()
    N = x.size(0)
    H = x.size(1)
    gate = torch.tanh if kind == "GNNPlus" else torch.sigmoid
    width = H if kind == "GNNPlus" else H // K
    periph = torch.zeros(N, K, width)
    if data.get("peripheral_edge_attr") is not None and "pew" in p:
        periph = periph + gate(p["pew"]) * feature_concat_encoder(sub(p, "peripheral_edge_embedding"),
                                                                  data["peripheral_edge_attr"]).sum(-2)
    if data.get("peripheral_configuration_attr") is not None and "pcw" in p:
        periph = periph + gate(p["pcw"]) * feature_concat_encoder(sub(p, "peripheral_configuration_embedding"),
                                                                  data["peripheral_configuration_attr"])
    num_graphs = int(batch[-1]) + 1
    vn = None
    if virtual_node:
        vn = F.embedding(torch.zeros(num_graphs, dtype=torch.long), p["virtualnode_embedding.weight"])

    def conv(l, h_in, k=None):
        if kind == "GNNPlus":
            lp = sub(p, f"gnns.{l}")
            return LO.kpginplus_forward(lp, h_in, edge_index, edge_attr[:, :k], None if pe_attr is None else pe_attr[:, :k - 1],
                                        periph[:, :k], K=k, combine_kind=combine_kind, training=training)
        if kind == "GNNPrime" and l >= num_l1_layer:
            return LO.gine_forward(sub(p, f"gins.{l - num_l1_layer}"), h_in, edge_index, edge_attr[:, :1], training=training)
        lp = sub(p, f"khop_gnns.{l}" if kind == "GNNPrime" else f"gnns.{l}")
        fwd = LO.kpgcn_forward if layer_kind == "KPGCN" else LO.kpgin_forward
        return fwd(lp, h_in, edge_index, edge_attr, pe_attr, periph, K=K, combine_kind=combine_kind)

    h_list, last_h = [x], x
    for l in range(num_layer):
        if virtual_node:
            h_list[l] = h_list[l] + vn[batch]
        if kind == "GNNPlus":
            k = min(l + 1, K)
            xs = torch.cat([h_list[j].unsqueeze(1) for j in range(l, l - k, -1)], dim=1)
            h = conv(l, xs, k)
        else:
            h = conv(l, h_list[l])
        h = batch_norm(p, f"norms.{l}.module", h, training)
        if residual:
            if kind == "GNNPlus":
                h = h + last_h
                last_h = h
            else:
                h = h + h_list[l]
        h_list.append(h)
        if virtual_node and l < num_layer - 1:
            tmp = global_add_pool(h_list[l], batch, num_graphs) + vn
            upd = _vn_mlp(p, f"mlp_virtualnode_list.{l}", tmp, training)
            vn = vn + upd if residual else upd
    if JK == "concat":
        rep = torch.cat(h_list, dim=1)
    elif JK == "last":
        rep = h_list[-1]
    elif JK == "sum":
        rep = torch.stack(h_list, 0).sum(0)
    elif JK == "max":
        rep = torch.stack(h_list, -1).max(-1).values
    else:
        raise NameError(JK)
    return F.relu(F.linear(rep, p["output_proj.0.weight"], p["output_proj.0.bias"]))


def graph_regression_forward(p, data, **body_kw):
    x = body_forward(sub(p, "embedding_model"), data, **body_kw)
    num_graphs = int(data["batch"][-1]) + 1
    pooled = global_add_pool(x, data["batch"], num_graphs)
    return F.linear(pooled, p["regressor.weight"], p["regressor.bias"]).squeeze()
