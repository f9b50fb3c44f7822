"""CPU, world_size 2, gloo: the data-parallel plumbing bench.py uses for N > 1 (flat gradient bucket, mean
all-reduce, replica broadcast, disjoint shard seeds, max-over-ranks timing)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kp_gnn_amd import dp
    torch.manual_seed(100 + rank)  # deliberately different init per rank: broadcast must fix it
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.ReLU(), torch.nn.Linear(5, 1))
    dp.broadcast_model(model)
    flat = dp.flatten_grads(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]   # each rank owns its shard
    for _ in range(3):
        flat.zero_()
        loss = (model(xs).squeeze() - ys).abs().mean()
        loss.backward()
        local = flat.clone()
        dp.allreduce_mean(flat, world)
        opt.step()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    mean_ok = torch.allclose(flat, sum(gathered) / world, atol=1e-7)
    params = torch.cat([p.detach().flatten() for p in model.parameters()])
    plist = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(plist, params)
    t = dp.max_over_ranks(1.0 + rank, torch.device("cpu"), world)
    if rank == 0:
        out.put({"mean_ok": bool(mean_ok), "replicas_equal": bool(torch.equal(plist[0], plist[1])),
                 "views": all(p.grad.data_ptr() >= flat.data_ptr() for p in model.parameters()),
                 "tmax": t, "seeds": [dp.shard_seed(r, 4, b, 64) for r in range(2) for b in range(4)]})
    dist.destroy_process_group()


def test_flat_bucket_allreduce_world2():
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = out.get()
    assert res["mean_ok"] and res["replicas_equal"] and res["views"]
    assert res["tmax"] == 2.0
    assert len(set(res["seeds"])) == 8 and min(b - a for a, b in zip(res["seeds"], res["seeds"][1:])) == 64


def test_flat_parameter_bucket_steps_like_per_parameter_adam():
    """dp.flatten_params / flatten_grads: Adam on the single flat (256-B aligned, zero-padded) bucket takes exactly the
    steps of Adam on the individual parameters, and the module keeps seeing its parameters through the views."""
    import copy
    import torch
    from kp_gnn_amd import dp
    torch.manual_seed(3)
    ref = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.Tanh(), torch.nn.Linear(13, 5), torch.nn.Linear(5, 1))
    mod = copy.deepcopy(ref)
    flat_g = dp.flatten_grads(mod)
    flat_p = dp.flatten_params(mod)
    flat_p.grad = flat_g
    assert all(p.data_ptr() % 256 == flat_p.data_ptr() % 256 for p in mod.parameters())
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-2)
    opt = torch.optim.Adam([flat_p], lr=1e-2)
    for step in range(4):
        x = torch.randn(32, 7)
        for m, o in ((ref, opt_ref), (mod, opt)):
            loss = m(x).pow(2).mean()
            if m is ref:
                o.zero_grad()
                loss.backward()
            else:
                params, views = dp.grad_views(m)
                grads = torch.autograd.grad(loss, params)
                torch._foreach_copy_(views, list(grads))
            o.step()
    for a, b in zip(ref.parameters(), mod.parameters()):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    assert set(ref.state_dict()) == set(mod.state_dict())


# ----------------------------------------------------------------------------- unequal shards on a real KP-GIN+ body
def _dp_setup():
    """Five synthetic molecules, their K = 3 pre-transform per graph, and a GNNPlus + GraphRegression state dict."""
    import argparse
    import numpy as np
    from kp_gnn_amd import body as B, khop_transform as KT
    from kp_gnn_amd.layers import make_gnn_layer
    K, L, H, G = 3, 3, 16, 5
    raw = KT.synth_molecules(G, seed0=40)
    node_ptr, edge_ptr, ei, ea, x = raw
    pre = (K, 50, 6, 3, 50, 50, "spd")
    pairs = []
    for g in range(G):
        out = KT.khop_batch(*[a for a in __import__("kp_gnn_amd.dp", fromlist=["select_graphs"]).select_graphs(node_ptr, edge_ptr, ei, ea, x, [g])[:4]],
                            *pre, num_threads=1)
        pairs.append(int((out["edge_attr"] != 0).sum()))
    ns = argparse.Namespace(model_name="KPGINPlus", hidden_size=H, K=K, num_layer=L, num_hop1_edge=3, max_pe_num=50,
                            combine="geometric", eps=0., train_eps=False, aggr="add")
    torch.manual_seed(11)
    gnn = B.GNNPlus(num_layer=L, gnn_layer=make_gnn_layer(ns), JK="concat", norm_type="Batch", init_emb=B.EmbeddingEncoder(21, H),
                    residual=True, virtual_node=False, use_rd=False, num_hop1_edge=3, max_edge_count=50, max_hop_num=6,
                    max_distance_count=50, drop_prob=0.0)
    model = B.GraphRegression(gnn, "sum")
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    for k in sd:          # non-trivial running statistics (the oracle runs the norms in eval mode: batch-independent)
        if k.endswith("running_mean"):
            sd[k] = torch.randn_like(sd[k]) * 0.1
        if k.endswith("running_var"):
            sd[k] = torch.rand_like(sd[k]) + 0.5
    y = torch.randn(G, generator=torch.Generator().manual_seed(5))
    return raw, pre, pairs, sd, y, (K, L)


def _oracle_grads(sd, raw, pre, idx, y, weight, KL):
    """Gradient of weight * mean_{g in idx} |score_g - y_g| w.r.t. every float parameter, by the CPU oracle."""
    from kp_gnn_amd import dp
    from kp_gnn_amd.batch import collate_khop
    from oracle import kp_model_oracle as MO
    K, L = KL
    b = collate_khop(*dp.select_graphs(*raw, idx), pre, y=y[idx], num_threads=1)
    names = sorted(k for k, v in sd.items() if v.is_floating_point() and "running" not in k and not k.endswith(".eps"))
    p = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in sd.items()}
    score = MO.graph_regression_forward(p, b.as_dict(), kind="GNNPlus", layer_kind="KPGINPlus", K=K, num_layer=L,
                                        combine_kind="geometric", JK="concat", residual=True, training=False)
    loss = (score.reshape(-1) - b.y.reshape(-1)).abs().mean() * weight
    loss.backward()
    return names, [p[k].grad if p[k].grad is not None else torch.zeros_like(p[k]) for k in names]


def _dp_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kp_gnn_amd import dp
    torch.set_num_threads(2)
    raw, pre, pairs, sd, y, KL = _dp_setup()
    shards = dp.partition_by_pairs(pairs, world)
    idx = shards[rank]
    w = dp.shard_loss_weight(len(idx), len(pairs))
    names, grads = _oracle_grads(sd, raw, pre, idx, y, w, KL)
    holder = torch.nn.Module()
    holder.ps = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros_like(g)) for g in grads])
    flat = dp.flatten_grads(holder)                      # the product's flat bucket: .grad views, 256-B aligned
    for prm, g in zip(holder.ps, grads):
        prm.grad.copy_(g)
    dp.allreduce_sum(flat, world)
    if rank == 0:
        out.put({"shards": shards, "pairs": pairs, "grads": [prm.grad.detach().numpy().copy() for prm in holder.ps],
                 "names": names})     # (numpy: pickled by value - a shared-memory tensor would vanish with this process)
    dist.destroy_process_group()


def test_unequal_shards_sum_to_the_global_batch_gradient():
    """Two ranks, graphs partitioned by active pairs (unequal graph counts), each rank's mean loss weighted by
    n_local / n_global, SUM all-reduce of the product's flat bucket: the result is the gradient of the mean loss over the
    GLOBAL batch computed in one process - the reference's semantics (train_ZINC.py:34-36,42,181-185).  The per-rank
    model is the CPU oracle of a real KP-GIN+ body (the product refuses CPU tensors); norms in eval mode, since batch
    statistics are per replica by design (as in the reference's DataParallel)."""
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, out), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    # (read before the joins: a put of this size blocks until it is read - but never wait forever for a rank that died
    #  before its put: poll with a deadline and watch the exit codes)
    import time
    deadline = time.time() + 180
    while out.empty():
        dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
        assert not dead, f"a rank exited with {dead} before reporting"
        assert time.time() < deadline, "no result from rank 0 within 180 s"
        time.sleep(0.05)
    res = out.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    shards, pairs = res["shards"], res["pairs"]
    assert sorted(shards[0] + shards[1]) == list(range(5)) and len(shards[0]) != len(shards[1])
    load = [sum(pairs[i] for i in sh) for sh in shards]
    assert max(load) - min(load) <= max(pairs)                      # LPT bound: balanced by pairs
    raw, pre, pairs2, sd, y, KL = _dp_setup()
    assert pairs2 == pairs
    names, ref = _oracle_grads(sd, raw, pre, list(range(5)), y, 1.0, KL)
    assert names == res["names"]
    for n, a, b in zip(names, res["grads"], ref):
        a = torch.from_numpy(a)
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6 * max(1.0, float(b.abs().max()))), (n, float((a - b).abs().max()))


def test_partition_by_pairs_balances_and_is_deterministic():
    from kp_gnn_amd import dp
    w = [500, 120, 130, 900, 40, 60, 700, 300]
    sh = dp.partition_by_pairs(w, 3)
    assert sorted(sum(sh, [])) == list(range(8)) and sh == dp.partition_by_pairs(w, 3)
    load = [sum(w[i] for i in s) for s in sh]
    assert max(load) - min(load) <= max(w) and all(s == sorted(s) for s in sh)
    assert dp.shard_loss_weight(3, 12) == 0.25
