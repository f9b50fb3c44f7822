"""ORACLE (test infrastructure, not product code): plain-PyTorch CPU restatement of KP-GNN's K-hop
message-passing layers, written op for op in the reference's own *materialised* [E,K,D] sequence
(embedding -> index_select -> add -> masked_fill -> index_add_), i.e. the CPU path the HIP kernels are
checked against and the `cpu_baseline` that bench.py times.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Every function is functional: `p` is a dict holding tensors under the reference's state_dict key
names, so golden state_dicts load verbatim and autograd gives the parameter grads.

Follows
    /root/reference/layers/KPGIN.py:86-121       kpgin_forward
    /root/reference/layers/KPGINplus.py:61-88    kpginplus_forward
    /root/reference/layers/KPGCN.py:11-25,80-126 kpgcn_forward, khop_degree
    /root/reference/layers/gine.py:49-59         gine_forward
    /root/reference/layers/KPGraphSAGE.py:71-98  kpgraphsage_forward
    /root/reference/layers/combine.py:22-27      attention_combine (nn.LSTM written out gate by gate)
    /root/reference/layers/combine.py:43-58      geometric_combine
    /root/reference/run_simulation.py:70-93      kgin_forward (mask-only variant)
PyG's MessagePassing.propagate (third party, PyG 2.1.0, absent from /root/reference and from this
image) is restated from its documented contract in `propagate_sum`: x_j = x.index_select(0, src);
sum-scatter of message(...) at dst with dim_size = N.

Parity pin: tests/golden/{layers,combine}.pt were produced by the reference's own layer files
(tests/golden/make_golden.py); tests/test_oracle_golden.py checks outputs, input grads and every
parameter grad of this restatement against them (fp32, rtol 1e-5 / atol 1e-6).
"""
import torch
import torch.nn.functional as F


def sub(p, prefix):
    """View of dict `p` restricted to keys under `prefix.` (prefix stripped)."""
    n = len(prefix) + 1
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix + ".")}


def propagate_sum(num_nodes, edge_index, msg):
    out = torch.zeros([num_nodes] + list(msg.shape[1:]), dtype=msg.dtype, device=msg.device)
    return out.index_add_(0, edge_index[1], msg)


def edge_code_embedding(p, edge_attr, K):
    """hop1_edge_emb on column 0, hopk_edge_emb on columns 1.., padding row 0 (KPGIN.py:90-98)."""
    e = F.embedding(edge_attr[:, :1], p["hop1_edge_emb.weight"], padding_idx=0)
    if K > 1:
        ek = F.embedding(edge_attr[:, 1:], p["hopk_edge_emb.weight"], padding_idx=0)
        e = torch.cat([e, ek], dim=-2)
    return e


def add_path_encoding(p, x, pe_attr, K):
    """x[:, 1:] += hopk_node_path_emb(pe_attr) (KPGIN.py:92-94) - functional, no input mutation."""
    if K > 1 and pe_attr is not None:
        pe = F.embedding(pe_attr, p["hopk_node_path_emb.weight"], padding_idx=0)
        x = torch.cat([x[:, :1], x[:, 1:] + pe], dim=1)
    return x


def masked_message(x_j, edge_emb, mask):
    m = x_j + edge_emb
    return m.masked_fill(mask.unsqueeze(-1) == 0, 0.)


# ------------------------------------------------------------------------------------------------ combine
def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of a 1-layer nn.LSTM(batch_first): gates i,f,g,o; returns [N,T,Hd]."""
    N, T, _ = x.shape
    Hd = w_hh.shape[1]
    h = x.new_zeros(N, Hd)
    c = x.new_zeros(N, Hd)
    outs = [None] * T
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        gates = x[:, t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
        i, f, g, o = gates.chunk(4, dim=-1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs, dim=1)


def attention_combine(p, x):
    """combine.py:22-27.  p keys: attention_lstm.{weight_ih_l0,weight_hh_l0,bias_ih_l0,bias_hh_l0}[_reverse]."""
    q = "attention_lstm."
    fw = lstm_direction(x, p[q + "weight_ih_l0"], p[q + "weight_hh_l0"], p[q + "bias_ih_l0"], p[q + "bias_hh_l0"], False)
    bw = lstm_direction(x, p[q + "weight_ih_l0_reverse"], p[q + "weight_hh_l0_reverse"],
                        p[q + "bias_ih_l0_reverse"], p[q + "bias_hh_l0_reverse"], True)
    score = torch.cat([fw, bw], dim=-1).sum(-1)  # N,K
    score = torch.softmax(score, dim=1).unsqueeze(-1)
    return (x * score).sum(1)


def geometric_thetas(alphas, K):
    """combine.py:48-58."""
    a = torch.sigmoid(alphas)
    th = torch.stack([a * (1 - a) ** i for i in range(K)], dim=0).unsqueeze(0)  # 1,K,D
    return torch.softmax(th, dim=-2)


def geometric_combine(p, x):
    return (x * geometric_thetas(p["alphas"], x.size(-2))).sum(-2)


def combine(p, x, K, kind):
    if K == 1:
        return torch.squeeze(x)  # KPGIN.py:64 (quirk: squeezes every unit dim)
    if kind == "attention":
        return attention_combine(sub(p, "combine"), x)
    if kind == "geometric":
        return geometric_combine(sub(p, "combine"), x)
    raise ValueError("Not implemented combine function")


# ------------------------------------------------------------------------------------------------ layers
def kpgin_forward(p, x, edge_index, edge_attr, pe_attr=None, peripheral_attr=None, *, K, combine_kind="geometric"):
    """layers/KPGIN.py:86-113."""
    dk = p["hop_proj1"].shape[1]
    x = x.reshape(-1, K, dk)
    x = add_path_encoding(p, x, pe_attr, K)
    e_emb = edge_code_embedding(p, edge_attr, K)
    x_j = x.index_select(0, edge_index[0])
    x_n = propagate_sum(x.size(0), edge_index, masked_message(x_j, e_emb, edge_attr))
    if peripheral_attr is not None:
        x_n = x_n + peripheral_attr
    x = x_n + (1 + p["eps"]) * x
    x = x.permute(1, 0, 2)
    x = F.relu(torch.matmul(x, p["hop_proj1"]) + p["hop_bias1"].unsqueeze(1))
    x = F.relu(torch.matmul(x, p["hop_proj2"]) + p["hop_bias2"].unsqueeze(1))
    x = x.permute(1, 0, 2)
    x = combine(p, x, K, combine_kind)
    if K > 1:
        x = F.linear(x, p["combine_proj.weight"], p["combine_proj.bias"])
    return x


def mlp_bn(p, prefix, x, training, momentum=0.1, eps=1e-5):
    """nn.Sequential(Linear, BatchNorm1d, ReLU, Linear, BatchNorm1d, ReLU) (KPGINplus.py:25-30).
    Running statistics in `p` are updated in place when training (as nn.BatchNorm1d does)."""
    q = prefix + "."
    x = F.linear(x, p[q + "0.weight"], p[q + "0.bias"])
    x = F.batch_norm(x, p.get(q + "1.running_mean"), p.get(q + "1.running_var"), p[q + "1.weight"], p[q + "1.bias"],
                     training, momentum, eps)
    x = F.relu(x)
    x = F.linear(x, p[q + "3.weight"], p[q + "3.bias"])
    x = F.batch_norm(x, p.get(q + "4.running_mean"), p.get(q + "4.running_var"), p[q + "4.weight"], p[q + "4.bias"],
                     training, momentum, eps)
    return F.relu(x)


def kpginplus_forward(p, x, edge_index, edge_attr, pe_attr=None, peripheral_attr=None, *, K,
                      combine_kind="geometric", training=True):
    """layers/KPGINplus.py:61-88.  x is [N,K,H]."""
    x = add_path_encoding(p, x, pe_attr, K)
    e_emb = edge_code_embedding(p, edge_attr, K)
    x_j = x.index_select(0, edge_index[0])
    x_n = F.gelu(propagate_sum(x.size(0), edge_index, masked_message(x_j, e_emb, edge_attr)))
    if peripheral_attr is not None:
        x_n = x_n + peripheral_attr
    return mlp_bn(p, "mlp", combine(p, x_n, K, combine_kind), training)


def khop_degree(index, num_nodes, index_mask):
    """layers/KPGCN.py:11-25."""
    out = torch.zeros((num_nodes, index_mask.size(-1)))
    one = (index_mask > 0).to(out.dtype)
    return out.scatter_add_(0, index.unsqueeze(-1).expand(-1, index_mask.size(-1)), one)


def kpgcn_forward(p, x, edge_index, edge_attr, pe_attr=None, peripheral_attr=None, *, K, combine_kind="geometric"):
    """layers/KPGCN.py:80-118."""
    N = x.size(0)
    loop = torch.arange(N, dtype=edge_index.dtype)
    edge_index = torch.cat([edge_index, loop.unsqueeze(0).repeat(2, 1)], dim=1)  # add_self_loops
    edge_attr = torch.cat([edge_attr, torch.ones([N, K], dtype=edge_attr.dtype)], dim=0)
    x = F.linear(x, p["hop_proj.weight"], p["hop_proj.bias"])
    dk = x.size(-1) // K
    x = x.view(-1, K, dk)
    x = add_path_encoding(p, x, pe_attr, K)
    e_emb = edge_code_embedding(p, edge_attr, K)
    row, col = edge_index
    deg = khop_degree(col, N, edge_attr)
    dis = deg.pow(-0.5)
    norm = dis[row] * dis[col]
    x_j = x.index_select(0, row)
    msg = (norm.unsqueeze(-1) * (x_j + e_emb)).masked_fill(edge_attr.unsqueeze(-1) == 0, 0.)
    x = F.relu(propagate_sum(N, edge_index, msg))
    if peripheral_attr is not None:
        x = x + peripheral_attr
    x = combine(p, x, K, combine_kind)
    if K > 1:
        x = F.linear(x, p["combine_proj.weight"], p["combine_proj.bias"])
    return x


def kpgraphsage_forward(p, x, edge_index, edge_attr, pe_attr=None, peripheral_attr=None, *, K, combine_kind="geometric"):
    """layers/KPGraphSAGE.py:71-98 (sum aggregation)."""
    dk = p["hop_proj"].shape[1] // 2
    x = x.reshape(-1, K, dk)
    x = add_path_encoding(p, x, pe_attr, K)
    e_emb = edge_code_embedding(p, edge_attr, K)
    x_j = x.index_select(0, edge_index[0])
    x_n = propagate_sum(x.size(0), edge_index, masked_message(x_j, e_emb, edge_attr))
    if peripheral_attr is not None:
        x_n = x_n + peripheral_attr
    h = torch.cat([x, x_n], dim=-1).permute(1, 0, 2)
    h = (torch.matmul(h, p["hop_proj"]) + p["hop_bias"].unsqueeze(1)).permute(1, 0, 2)
    h = F.normalize(F.relu(h), p=2, dim=-1)
    h = combine(p, h, K, combine_kind)
    if K > 1:
        h = F.linear(h, p["combine_proj.weight"], p["combine_proj.bias"])
    return h


def gine_forward(p, x, edge_index, edge_attr, *, training=True):
    """layers/gine.py:49-59.  edge_attr is the [E,1] hop-1 column of the K-hop edge list."""
    H = p["hop1_edge_emb.weight"].shape[1]
    x = x.view(-1, 1, H)
    e_emb = F.embedding(edge_attr, p["hop1_edge_emb.weight"], padding_idx=0)
    x_j = x.index_select(0, edge_index[0])
    out = propagate_sum(x.size(0), edge_index, masked_message(x_j, e_emb, edge_attr))
    out = out + (1 + p["eps"]) * x
    return mlp_bn(p, "mlp", out.squeeze(), training)


def kgin_forward(p, x, edge_index, edge_attr, *, K, batch=None):
    """run_simulation.py:70-85: mask-only K-hop GIN; `batch` given = the script's `args.graph` sum pooling (:83-84)."""
    hs = p["hop_proj1"].shape[1]
    x = F.linear(x, p["proj.weight"], p["proj.bias"]).view(-1, K, hs)
    x_j = x.index_select(0, edge_index[0])
    x_n = propagate_sum(x.size(0), edge_index, x_j.masked_fill(edge_attr.unsqueeze(-1) == 0, 0.))
    x = x_n + (1 + p["eps"]) * x
    x = x.permute(1, 0, 2)
    x = F.relu(torch.matmul(x, p["hop_proj1"]) + p["hop_bias1"].unsqueeze(1))
    x = F.relu(torch.matmul(x, p["hop_proj2"]) + p["hop_bias2"].unsqueeze(1))
    x = x.permute(1, 0, 2).contiguous().view(-1, K * hs)
    x = F.linear(x, p["combine_proj.weight"], p["combine_proj.bias"])
    if batch is not None:
        x = x.new_zeros(int(batch[-1]) + 1, x.size(1)).index_add_(0, batch, x)
    return x
