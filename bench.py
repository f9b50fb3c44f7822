#!/usr/bin/env python3
"""bench.py - graphs/s of the KP-GIN+ training step (fwd + bwd + Adam) on ZINC-12k-shaped synthetic batches.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--model KPGINPlus|KPGIN] [--combine geometric|attention]

Workload (BASELINE.json configs[1], reference README.md:127 `train_ZINC.py --residual --K=8 --model_name=KPGINPlus
--num_layer=8 --hidden_size=104`): GNNPlus body, K=8, L=8, h=104, geometric combine, JK=concat, BatchNorm, sum
pooling, L1 loss, Adam; synthetic molecule graphs of ZINC-12k's shape (kpgnn_host.h, ~23 nodes / ~500 K=8-spd
edges per graph), exact K-hop pre-transform by libkpgnn_host.so, random-init weights (seed 0).
A "step" is one optimisation step on one pre-staged batch of B graphs per GPU (inputs and the K-hop CSR
resident in HBM before the timed region).  N > 1: one process per GPU (torchrun), graphs sharded across
ranks, model replicated, one RCCL all-reduce (mean) of the flat gradient bucket per step; weak scaling.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline":     HIP-event timing of the aggregation kernel launches inside the timed region vs the
                  algorithmic bytes of SURVEY.md 8(d) (HBM bound, peak 8 TB/s);
  "cpu_baseline": the oracle ("port": oracle/kp_model_oracle.py, the reference's materialised [E,K,D] op
                  sequence in plain PyTorch) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

dp = None
ops_dense = None
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable
T_START = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:6.1f}s] {msg}", file=sys.stderr, flush=True)


from kp_gnn_amd._env import usable_cpus  # noqa: E402


def build_model(args, device):
    from kp_gnn_amd import body as B
    from kp_gnn_amd.layers import make_gnn_layer
    ns = argparse.Namespace(model_name=args.model, hidden_size=args.hidden, K=args.K, num_layer=args.layers,
                            num_hop1_edge=3, max_pe_num=50, combine=args.combine, eps=0., train_eps=False, aggr="add")
    torch.manual_seed(0)
    gnn = B.make_GNN(ns)(num_layer=args.layers, gnn_layer=make_gnn_layer(ns), JK="concat", norm_type="Batch",
                         init_emb=B.EmbeddingEncoder(21, args.hidden), residual=True, virtual_node=False, use_rd=False,
                         num_hop1_edge=3, max_edge_count=50, max_hop_num=6, max_distance_count=50,
                         wo_peripheral_edge=False, wo_peripheral_configuration=False, drop_prob=0.0)
    model = B.GraphRegression(gnn, "sum")
    return model.to(device).train()


def fwd_bwd(model, batch, flat_grad):
    """Forward + backward; the parameter gradients land in the flat bucket with ONE multi-tensor copy (autograd.grad
    returns them instead of running ~190 per-parameter accumulate kernels into pre-zeroed .grad views)."""
    score = model(batch)
    loss = (score.squeeze() - batch.y.squeeze()).abs().mean()  # train_ZINC.py:42
    params, views = dp.grad_views(model)
    grads = torch.autograd.grad(loss, params, allow_unused=True)
    used = [(v, g) for v, g in zip(views, grads) if g is not None]
    torch._foreach_copy_([v for v, _ in used], [g for _, g in used])
    # (parameters without a gradient - e.g. the never-trained path-encoding tables, Q1 - keep the zeros the flat bucket
    #  was created with: nothing ever writes their views)
    return loss


def train_step(model, batch, opt, flat_grad, world, graph=None):
    """One optimisation step.  With `graph` (a captured hipGraph of fwd+bwd on this batch's static tensors) the
    ~1,500 launches of forward+backward replay as one graph launch; the gradient all-reduce and the fused Adam
    step stay eager (a collective inside a captured graph is the one thing that cannot be rehearsed on 1 GPU)."""
    trace = os.environ.get("KPGNN_BENCH_TRACE") == "1"
    if trace:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    if graph is not None:
        graph[0].replay()
        loss = graph[1]
    else:
        loss = fwd_bwd(model, batch, flat_grad)
    if trace:
        torch.cuda.synchronize(); t1 = time.perf_counter()
    dp.allreduce_mean(flat_grad, world)
    if trace:
        torch.cuda.synchronize(); t2 = time.perf_counter()
    opt.step()
    if trace:
        torch.cuda.synchronize(); t3 = time.perf_counter()
        log(f"step phases: fwd+bwd {1e3*(t1-t0):.1f} ms, all-reduce {1e3*(t2-t1):.1f} ms, optimizer {1e3*(t3-t2):.1f} ms")
    return loss


def capture_graphs(model, batches, flat_grad):
    """hipGraph capture of fwd+bwd, one graph per pre-staged batch (shapes differ between batches)."""
    graphs = []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for b in batches:                      # warm every batch on the side stream (allocator, caches)
            fwd_bwd(model, b, flat_grad)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    pool = None
    # with a process group alive, its watchdog thread may query events while this thread captures: thread-local capture
    # mode keeps those calls legal (the captured region itself holds no collective)
    mode = "thread_local" if (torch.distributed.is_available() and torch.distributed.is_initialized()) else "global"
    for b in batches:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool, capture_error_mode=mode):
            loss = fwd_bwd(model, b, flat_grad)
        pool = g.pool()
        graphs.append((g, loss))
    return graphs


def pmc_traffic(args, kernel):
    """HBM bytes per launch of `kernel` from the PMC counters.  A process cannot profile itself, so the figure comes
    from the committed rocprofv3 --pmc passes of this same command (profiles/r01/pmc_traffic.json says how they were
    collected and corrected) and is only reported when the workload is the one that was profiled; otherwise null."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "pmc_traffic.json")
    key = f"{args.model}|B{args.batch}|K{args.K}|L{args.layers}|h{args.hidden}|{args.combine}"
    try:
        with open(path) as fh:
            j = json.load(fh)
        if j.get("workload_key") == key and kernel in j.get("kernels", {}):
            return {"traffic": j["kernels"][kernel]["traffic_bytes_per_launch"],
                    "traffic_source": "profiles/r01/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"}
    except (OSError, ValueError):
        pass
    return {}


def cpu_baseline(args, state_dict, threads):
    """Oracle (CPU restatement of the reference path) fwd+bwd on a bounded sample of the same workload."""
    from kp_gnn_amd.batch import synthetic_zinc_batch
    from oracle import kp_model_oracle as MO
    torch.set_num_threads(threads)
    nb = args.cpu_graphs
    b = synthetic_zinc_batch(nb, seed0=10_000_000, K=args.K)
    data = b.as_dict()
    kind, layer_kind = {"KPGINPlus": ("GNNPlus", "KPGINPlus"), "KPGIN": ("GNN", "KPGIN")}[args.model]
    p = {}
    for k, v in state_dict.items():
        v = v.detach().cpu().clone()
        p[k] = v.requires_grad_(True) if (v.is_floating_point() and "running" not in k and not k.endswith(".eps")) else v
    times = []
    for it in range(1 + args.cpu_iters):
        for v in p.values():
            if v.requires_grad:
                v.grad = None
        t0 = time.perf_counter()
        score = MO.graph_regression_forward(p, data, kind=kind, layer_kind=layer_kind, K=args.K, num_layer=args.layers,
                                            combine_kind=args.combine, JK="concat", residual=True, training=True)
        loss = (score.squeeze() - data["y"].squeeze()).abs().mean()
        loss.backward()
        dt = time.perf_counter() - t0
        log(f"  cpu iter {it}: {dt:.2f}s")
        if it > 0:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(nb / med, 2), "unit": "graphs/s", "cores": threads, "kind": "port",
            "sample": f"{nb} graphs x {args.cpu_iters} fwd+bwd iterations (median), oracle/kp_model_oracle.py, "
                      f"torch {torch.__version__} CPU, no optimizer step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2048, help="graphs per GPU per step")
    ap.add_argument("--num-batches", type=int, default=4, help="distinct pre-staged batches cycled through")
    ap.add_argument("--model", default="KPGINPlus", choices=("KPGINPlus", "KPGIN"))
    ap.add_argument("--combine", default="geometric", choices=("geometric", "attention"))
    ap.add_argument("--K", type=int, default=8)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--hidden", type=int, default=104)
    ap.add_argument("--cpu-graphs", type=int, default=128)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-launch HIP-event timing")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly (no hipGraph replay)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="collective backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse on one GPU)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: all ranks use cuda:0")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    device = torch.device("cuda", 0 if args.share_device else local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)
        else:
            torch.distributed.init_process_group("gloo")

    from kp_gnn_amd import ops
    from kp_gnn_amd.batch import synthetic_zinc_batch
    global dp, ops_dense
    from kp_gnn_amd import dp, ops_dense

    threads = max(1, usable_cpus() // max(1, min(world, 8)))
    torch.set_num_threads(threads)
    if rank == 0:
        log(f"world={world} host threads/rank={threads} (os.cpu_count={os.cpu_count()})")
    t_data = time.perf_counter()
    batches = []
    for i in range(args.num_batches):  # each rank owns its shard of graphs (distinct seeds)
        seed0 = dp.shard_seed(rank, args.num_batches, i, args.batch)
        b = synthetic_zinc_batch(args.batch, seed0=seed0, K=args.K, num_threads=threads).to(device)
        b.build_csr()
        batches.append(b)
    torch.cuda.synchronize()
    t_data = time.perf_counter() - t_data
    if rank == 0:
        log(f"{args.num_batches} batches x {args.batch} graphs built + CSR on device in {t_data:.1f}s "
            f"(N={batches[0].num_nodes}, E_khop={batches[0].edge_index.shape[1]})")

    model = build_model(args, device)
    if world > 1:  # identical replicas
        dp.broadcast_model(model)
    flat_grad = dp.flatten_grads(model)
    flat_param = dp.flatten_params(model)     # parameters and gradients: one flat bucket each (same order)
    flat_param.grad = flat_grad
    opt = torch.optim.Adam([flat_param], lr=1e-3, fused=True, capturable=True)   # elementwise: identical to per-parameter Adam

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        train_step(model, batches[i % len(batches)], opt, flat_grad, world)
    torch.cuda.synchronize()
    graphs = None
    if not args.no_graph:
        # (a capture failure is an error, not a silent downgrade to eager launches: --no-graph asks for those)
        graphs = capture_graphs(model, batches, flat_grad)
        for i in range(len(batches)):  # one replayed step per graph before timing
            train_step(model, batches[i], opt, flat_grad, world, graphs[i])
        torch.cuda.synchronize()
    if rank == 0:
        log(f"{args.warmup} warm-up steps done; launch mode: {'hipGraph replay' if graphs else 'eager'}")
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        j = i % len(batches)
        loss = train_step(model, batches[j], opt, flat_grad, world, graphs[j] if graphs else None)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = dp.max_over_ranks(elapsed, device, world)
    final_loss = float(loss.item())
    if rank == 0:
        log(f"timed {args.steps} steps in {elapsed:.3f}s")
    # Per-launch kernel timing (roofline leg): HIP events around every aggregation launch, on the launch
    # stream, over the same K steps run eagerly (events cannot bracket single kernels inside a replayed graph).
    timer = None
    if not args.no_roofline and rank == 0:
        timer = ops.LaunchTimer()
        ops.set_launch_timer(timer)
        train_step(model, batches[0], opt, flat_grad, 1)  # primes the byte-accounting caches
        timer.records.clear()
        for i in range(args.steps):
            train_step(model, batches[i % len(batches)], opt, flat_grad, 1)
        torch.cuda.synchronize()
        ops.set_launch_timer(None)

    if rank == 0:
        total_graphs = args.batch * world * args.steps
        b0 = batches[0]
        out = {
            "metric": "graphs/sec KP-GIN fwd+bwd, ZINC-12k K=8 L=8 h=104",
            "value": round(total_graphs / elapsed, 1),
            "unit": "graphs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ZINC-12k-shaped synthetic molecules, {args.model} K={args.K} L={args.layers} "
                                   f"h={args.hidden} {args.combine} combine, fwd+bwd+Adam, L1 loss",
                       "graphs_per_gpu_per_step": args.batch, "global_batch": args.batch * world,
                       "nodes_per_batch": b0.num_nodes, "khop_edges_per_batch": int(b0.edge_index.shape[1]),
                       "parallelism": f"dp{world}", "launch": "hipGraph replay of fwd+bwd" if graphs else "eager",
                       "final_loss": round(final_loss, 5),
                       "data_build_s": round(t_data, 2)},
        }
        if timer is not None:
            s = timer.summary()
            f = s.get("agg_fwd")
            if f:
                out["roofline"] = {"bound": "hbm", "kernel": "agg_fwd_kernel",
                                   "timing": "HIP events per launch, K eager steps after the timed region",
                                   "achieved": round(f["gbps"], 1),
                                   "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(f["gbps"] / HBM_PEAK_GBPS, 4),
                                   "traffic": None, "launches": f["launches"], "avg_launch_ms": round(f["avg_ms"], 4),
                                   "algorithmic_bytes_per_launch": int(f["bytes_per_launch"])}
                out["roofline"].update(pmc_traffic(args, "agg_fwd_kernel"))
            out["kernels"] = {k: {"launches": v["launches"], "avg_launch_ms": round(v["avg_ms"], 4),
                                  "algorithmic_GBps": round(v["gbps"], 1)} for k, v in s.items()}
            g = s.get("agg_bwd")
            if g:
                out["roofline_bwd"] = {"bound": "hbm", "kernel": "agg_bwd_kernel", "achieved": round(g["gbps"], 1),
                                       "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(g["gbps"] / HBM_PEAK_GBPS, 4),
                                       "launches": g["launches"], "avg_launch_ms": round(g["avg_ms"], 4),
                                       "algorithmic_bytes_per_launch": int(g["bytes_per_launch"])}
        if world == 1 and not args.no_cpu_baseline:
            sd = {k: v for k, v in model.state_dict().items()}
            log(f"cpu baseline: oracle on {threads} host threads, {args.cpu_graphs} graphs ...")
            out["cpu_baseline"] = cpu_baseline(args, sd, threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()        # rank 0 ran the per-launch timing leg alone: tear the communicator down together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
