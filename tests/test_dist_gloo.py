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
