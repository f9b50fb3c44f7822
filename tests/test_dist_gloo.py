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
