#!/usr/bin/env python3
"""bench.py - throughput of the K-hop message-passing hot path on synthetic batches of the reference's shapes.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload zinc|qm9|regular|zinc_gd16] [--batch B]
                    [--model KPGINPlus|KPGIN] [--combine geometric|attention] [--dense-peripheral]

Default workload = BASELINE.json configs[1] (reference README.md:127 `train_ZINC.py --residual --K=8
--model_name=KPGINPlus --num_layer=8 --hidden_size=104`): GNNPlus body, K=8, L=8, h=104, geometric combine, JK=concat,
BatchNorm, sum pooling, L1 loss, Adam; synthetic molecule graphs of ZINC-12k's shape (kpgnn_host.h, ~23 nodes / ~500
K=8-spd edges per graph), exact K-hop pre-transform by libkpgnn_host.so, random-init weights (seed 0).
Other workloads (the configurations of BASELINE.json that are parity cases for the default line):
  qm9        configs[2]: QM9-shaped molecules, KP-GIN (GNN body) K=6 L=8 h=120 spd, MSE loss, Adam (train_qm9.py:96,141-158)
  regular    configs[3]: random 3-regular graphs n=1280, run_simulation.py's KGINConv(16, K=8), FORWARD ONLY (eval, no grad),
             --batch graphs per step (the script uses 1; --batch 100 = its whole set of N=100 graphs at once)
  zinc_gd16  configs[4]: KP-GIN' (GNNPrime: one KP-GIN layer + 16 GINE layers) K=16 L=17 h=96 kernel=gd, L1, Adam
A "step" is one optimisation step (forward only for `regular`) on one pre-staged batch of B graphs per GPU (inputs and the
K-hop CSR resident in HBM before the timed region).  --fresh-batches: the reference's epoch instead (train_ZINC.py:224: a
shuffled DataLoader, no batch is seen twice) - a dataset of --dataset-graphs pre-transformed graphs stays resident in HBM
(kp_gnn_amd/dataset.py) and EVERY timed step first collates a new random subset of it (kpgnn_collate, inside the timed region).  N > 1: one process per GPU (torchrun), graphs sharded across
ranks, model replicated, one RCCL all-reduce of the flat gradient bucket per step; weak scaling (--batch graphs per GPU).
--scaling strong: --batch is ONE global batch, partitioned over the ranks by active pairs (dp.partition_by_pairs), each rank's
mean loss weighted by its share of the graphs (dp.shard_loss_weight), gradients summed.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline":     HIP-event timing of the aggregation kernel launches vs the algorithmic bytes of SURVEY.md 8(d)
                  (HBM bound, peak 8 TB/s), with the workload's own A and D;
  "cpu_baseline": the oracle ("port": oracle/, the reference's materialised [E,K,D] op sequence in plain PyTorch)
                  timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import torch

dp = None
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable
T_START = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:6.1f}s] {msg}", file=sys.stderr, flush=True)


from kp_gnn_amd._env import usable_cpus  # noqa: E402

# workload -> defaults.  metric: the JSON's "metric" string (the default line carries BASELINE.json's).
WORKLOADS = {
    "zinc": dict(model="KPGINPlus", K=8, layers=8, hidden=104, batch=2048, kernel="spd", loss="l1", train=True,
                 metric="graphs/sec KP-GIN fwd+bwd, ZINC-12k K=8 L=8 h=104"),
    "qm9": dict(model="KPGIN", K=6, layers=8, hidden=120, batch=128, kernel="spd", loss="mse", train=True,
                metric="graphs/sec KP-GIN fwd+bwd, QM9-shaped K=6 L=8 h=120"),
    "regular": dict(model="KGIN", K=8, layers=1, hidden=16, batch=1, kernel="spd", loss=None, train=False,
                    metric="graphs/sec KGINConv forward, 3-regular n=1280 K=8 h=16 (run_simulation.py)"),
    "zinc_gd16": dict(model="KPGINPrime", K=16, layers=17, hidden=96, batch=2048, kernel="gd", loss="l1", train=True,
                      metric="graphs/sec KP-GIN' fwd+bwd, ZINC-12k K=16 L=17 h=96 gd"),
}


def make_batch(args, seed0, threads):
    from kp_gnn_amd import batch as KB
    if args.workload == "qm9":
        return KB.synthetic_qm9_batch(args.batch, seed0=seed0, K=args.K, kernel=args.kernel, num_threads=threads)
    if args.workload == "regular":
        return KB.synthetic_regular_batch(args.batch, seed0=seed0, n=1280, degree=3, K=args.K, num_threads=threads)
    return KB.synthetic_zinc_batch(args.batch, seed0=seed0, K=args.K, kernel=args.kernel, num_threads=threads)


def build_model(args, device):
    from kp_gnn_amd import body as B
    from kp_gnn_amd.layers import KGINConv, make_gnn_layer
    torch.manual_seed(0)
    if args.workload == "regular":
        return KGINConv(args.hidden, args.K).to(device).eval()          # run_simulation.py:104-107,133
    qm9 = args.workload == "qm9"
    ns = argparse.Namespace(model_name=args.model, hidden_size=args.hidden, K=args.K, num_layer=args.layers,
                            num_hop1_edge=4 if qm9 else 3, max_pe_num=50, combine=args.combine, eps=0., train_eps=False,
                            aggr="add")
    enc = B.QM9InputEncoder(args.hidden) if qm9 else B.EmbeddingEncoder(21, args.hidden)
    kw = dict(max_edge_count=20, max_hop_num=5, max_distance_count=15) if qm9 else \
        dict(max_edge_count=50, max_hop_num=6, max_distance_count=50)
    gnn = B.make_GNN(ns)(num_layer=args.layers, gnn_layer=make_gnn_layer(ns), JK="concat", norm_type="Batch",
                         init_emb=enc, residual=not qm9, virtual_node=False, use_rd=False, num_hop1_edge=ns.num_hop1_edge,
                         wo_peripheral_edge=False, wo_peripheral_configuration=False, drop_prob=0.0, **kw)
    model = B.GraphRegression(gnn, "sum")
    return model.to(device).train()


def loss_of(args, score, y):
    from kp_gnn_amd.ops_dense import regression_loss
    return regression_loss(score, y, args.loss)                          # train_ZINC.py:42 / train_qm9.py:96


def fwd_bwd(args, model, batch, flat_grad):
    """Forward + backward; the parameter gradients land in the flat bucket with ONE multi-tensor copy (autograd.grad
    returns them instead of running ~190 per-parameter accumulate kernels into pre-zeroed .grad views).
    Forward-only workloads (`regular`) just run the layer as the reference's script does (eval, no grad)."""
    if not args.train:
        with torch.no_grad():
            return model(batch.x, batch.edge_index, batch.edge_attr, batch.batch)
    from kp_gnn_amd.ops_dense import regression_loss_and_grad
    score = model(batch)
    loss, dscore = regression_loss_and_grad(score, batch.y, args.loss)   # train_ZINC.py:42 / train_qm9.py:96, with its gradient
    w = getattr(batch, "_loss_weight", None)
    if w is not None:      # --scaling strong: this rank's share n_r / G of the global mean loss (dp.shard_loss_weight)
        loss, dscore = loss * w, dscore * w
    params, views = dp.grad_views(model)
    from kp_gnn_amd import ops
    with ops.deferred_reductions():      # (gradients are read after the block: the weight-gradient reduces ride along)
        grads = torch.autograd.grad(score, params, grad_outputs=dscore, allow_unused=True)
    used = [(v, g) for v, g in zip(views, grads) if g is not None]
    dp.copy_grads([v for v, _ in used], [g for _, g in used])
    # (parameters without a gradient - e.g. the never-trained path-encoding tables, Q1 - keep the zeros the flat bucket
    #  was created with: nothing ever writes their views)
    return loss


def train_step(args, model, batch, opt, flat_grad, world, graph=None):
    """One step.  With `graph` (a captured hipGraph of fwd+bwd on this batch's static tensors) the launches of
    forward+backward replay as one graph launch; the gradient all-reduce and the Adam step stay eager (a collective inside a
    captured graph is the one thing that cannot be rehearsed on 1 GPU).  --adam-in-graph (1 GPU) makes the Adam launch - its
    step number on the device - the graph's last node instead."""
    stepped = False
    if graph is not None:
        graph[0].replay()
        out, stepped = graph[1], graph[2]
    else:
        out = fwd_bwd(args, model, batch, flat_grad)
    if args.train and not stepped:
        if args.scaling == "strong":
            dp.allreduce_sum(flat_grad, world)      # (unequal shards, losses pre-weighted: the sum IS the global-mean gradient)
        else:
            dp.allreduce_mean(flat_grad, world)
        opt.step()
    return out


def capture_graphs(args, model, batches, flat_grad, opt=None):
    """hipGraph capture of fwd+bwd (+ the optimiser step when `opt` is given: single process, device-side step number), one
    graph per pre-staged batch (shapes differ between batches)."""
    graphs = []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for b in batches:                      # warm every batch on the side stream (allocator, caches)
            fwd_bwd(args, model, b, flat_grad)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    pool = None
    # with a process group alive, its watchdog thread may query events while this thread captures: thread-local capture
    # mode keeps those calls legal (the captured region itself holds no collective)
    mode = "thread_local" if (torch.distributed.is_available() and torch.distributed.is_initialized()) else "global"
    for b in batches:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool, capture_error_mode=mode):
            out = fwd_bwd(args, model, b, flat_grad)
            if opt is not None:
                opt.step()
        pool = g.pool()
        graphs.append((g, out, opt is not None))
    return graphs


def csrc_digest():
    """Hash of the kernel sources: a committed PMC traffic figure is only reported next to timings of the SAME kernels."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "kp_gnn_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def stream_copy_gbps(device):
    """What a plain device-to-device copy of a [N,8,104]-sized fp32 tensor reaches on this GPU (read + write bytes over HIP-event
    time): the practical ceiling next to the nominal 8 TB/s of the roofline."""
    x = torch.empty(47450 * 8 * 104, dtype=torch.float32, device=device).normal_()
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    return 2 * x.numel() * 4 * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def traffic_key(args):
    """Every flag that changes which kernels run (or how) is part of the key of a committed PMC traffic figure."""
    key = f"{args.workload}|{args.model}|B{args.batch}|K{args.K}|L{args.layers}|h{args.hidden}|{args.combine}"
    for flag, tag in ((args.dtype != "f32", args.dtype), (args.dense_peripheral, "dense-peripheral"), (args.fresh_batches, "fresh-batches")):
        if flag:
            key += "|" + tag
    return key


def pmc_traffic(args, kernel):
    """HBM bytes per launch of `kernel` from the PMC counters.  A process cannot profile itself, so the figure comes
    from the committed rocprofv3 --pmc passes of this same command (profiles/r03/pmc_traffic.json says how they were
    collected and corrected).  It is only reported when the workload AND the kernel sources (csrc digest) are the ones
    that were profiled; otherwise null - a stale figure next to fresh timings would be worse than none."""
    path = os.path.join(ROOT, "profiles", "r03", "pmc_traffic.json")
    key = traffic_key(args)
    try:
        with open(path) as fh:
            j = json.load(fh)
        if j.get("workload_key") == key and j.get("csrc_digest") == csrc_digest() and kernel in j.get("kernels", {}):
            return {"traffic": j["kernels"][kernel]["traffic_bytes_per_launch"],
                    "traffic_source": "profiles/r03/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                      "this command on these kernel sources)"}
    except (OSError, ValueError):
        pass
    return {}


def cpu_baseline(args, state_dict, threads):
    """Oracle (CPU restatement of the reference path) on a bounded sample of the same workload."""
    from oracle import kp_layers_oracle as LO
    from oracle import kp_model_oracle as MO
    torch.set_num_threads(threads)
    saved = args.batch
    args.batch = nb = min(args.cpu_graphs, saved) if args.workload != "regular" else 1
    b = make_batch(args, 10_000_000, threads)
    args.batch = saved
    data = b.as_dict()
    p = {}
    for k, v in state_dict.items():
        v = v.detach().cpu().clone()
        p[k] = v.requires_grad_(True) if (args.train and v.is_floating_point() and "running" not in k and not k.endswith(".eps")) else v
    kind, layer_kind = {"KPGINPlus": ("GNNPlus", "KPGINPlus"), "KPGIN": ("GNN", "KPGIN"), "KPGINPrime": ("GNNPrime", "KPGIN"),
                        "KGIN": (None, None)}[args.model]
    times = []
    for it in range(1 + args.cpu_iters):
        for v in p.values():
            if v.requires_grad:
                v.grad = None
        t0 = time.perf_counter()
        if args.workload == "regular":
            with torch.no_grad():
                LO.kgin_forward(p, data["x"], data["edge_index"], data["edge_attr"], K=args.K)
        else:
            score = MO.graph_regression_forward(p, data, kind=kind, layer_kind=layer_kind, K=args.K, num_layer=args.layers,
                                                combine_kind=args.combine, JK="concat", residual=args.workload != "qm9",
                                                training=True)
            loss_of(args, score, data["y"]).backward()
        dt = time.perf_counter() - t0
        log(f"  cpu iter {it}: {dt:.2f}s")
        if it > 0:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    what = "forward (no grad)" if not args.train else "fwd+bwd, no optimizer step"
    return {"value": round(nb / med, 2), "unit": "graphs/s", "cores": threads, "kind": "port",
            "sample": f"{nb} graphs x {args.cpu_iters} {what} iterations (median), oracle/, torch {torch.__version__} CPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="zinc", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="graphs per GPU per step (default: the workload's)")
    ap.add_argument("--num-batches", type=int, default=4, help="distinct pre-staged batches cycled through")
    ap.add_argument("--model", default=None, choices=("KPGINPlus", "KPGIN", "KPGINPrime"))
    ap.add_argument("--combine", default="geometric", choices=("geometric", "attention"))
    ap.add_argument("--K", type=int, default=None)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--hidden", type=int, default=None)
    ap.add_argument("--dtype", default="f32", choices=("f32", "bf16"),
                    help="storage of the K-hop streams (hop-slot rows, saved S, dL/dS): bf16 = 2-byte rows with fp32 sums "
                         "(KPGNN_STORE_BF16, KP-GIN+ path); every parameter, state and accumulator stays fp32")
    ap.add_argument("--dense-peripheral", action="store_true",
                    help="hand the layers the dense [N,K,D] peripheral tensor instead of its dictionary form")
    ap.add_argument("--cpu-graphs", type=int, default=128)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-launch HIP-event timing")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly (no hipGraph replay)")
    ap.add_argument("--adam-in-graph", action="store_true",
                    help="1 GPU: capture the optimiser step (device-side step number) in the graph; measured 20 us SLOWER per step "
                         "than the eager launch behind the replay (its 480 blocks take a same-address ticket each)")
    ap.add_argument("--fresh-batches", action="store_true",
                    help="every step collates a NEW random subset of a resident dataset (shuffled epochs, drop_last), collate "
                         "inside the timed region; launches are eager (batch shapes differ from step to step)")
    ap.add_argument("--dataset-graphs", type=int, default=10000, help="--fresh-batches: graphs per rank in the resident dataset "
                    "(10,000 = ZINC-12k's training split)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak: --batch graphs PER GPU (the default line).  strong: --batch is ONE global batch, partitioned over the "
                         "ranks by active (edge, hop) pairs (dp.partition_by_pairs), every rank's mean loss weighted by its share "
                         "(dp.shard_loss_weight) and the gradients summed - the reference's DataParallel step on a fixed batch")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="collective backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse on one GPU)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: all ranks use cuda:0")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    for k in ("model", "K", "layers", "hidden", "batch"):
        if getattr(args, k) is None:
            setattr(args, k, wl[k])
    args.kernel, args.loss, args.train = wl["kernel"], wl["loss"], wl["train"]
    if args.workload == "regular":
        args.model = "KGIN"
    if args.scaling == "strong" and (args.fresh_batches or not args.train):
        raise SystemExit("--scaling strong partitions pre-staged global batches of a training workload")

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

    from kp_gnn_amd import body as B
    from kp_gnn_amd import ops
    global dp
    from kp_gnn_amd import dp
    if args.dense_peripheral:
        B.MAX_DICT_ROWS = 0
    if args.dtype == "bf16":
        if args.model != "KPGINPlus" or args.combine != "geometric" or args.dense_peripheral or args.hidden % 8:
            raise SystemExit("--dtype bf16: the bf16-storage kernels exist for KPGINPlus + geometric combine + dictionary "
                             "peripheral features with hidden % 8 == 0")
        ops.set_storage_dtype(torch.bfloat16)

    threads = max(1, usable_cpus() // max(1, min(world, 8)))
    torch.set_num_threads(threads)
    if rank == 0:
        log(f"workload={args.workload} world={world} host threads/rank={threads} (os.cpu_count={os.cpu_count()})")
    t_data = time.perf_counter()
    batches = []
    dataset = sampler = None
    if args.fresh_batches:
        # the whole (per-rank) dataset: host pre-transform once, CSR built once on the device, resident from here on
        import numpy as np
        from kp_gnn_amd.dataset import KHopDataset
        if args.dataset_graphs < args.batch:
            raise SystemExit("--dataset-graphs must be >= --batch")
        saved, args.batch = args.batch, args.dataset_graphs
        host = make_batch(args, dp.shard_seed(rank, 1, 0, args.dataset_graphs), threads)
        args.batch = saved
        dataset = KHopDataset.from_collated(host, host.node_ptr, device)
        del host
        rng = np.random.default_rng(1234 + rank)

        def sampler(state={"perm": None, "pos": 0}):
            """Shuffled epochs with drop_last, as DataLoader(shuffle=True) would (train_ZINC.py:224)."""
            if state["perm"] is None or state["pos"] + args.batch > dataset.G:
                state["perm"], state["pos"] = rng.permutation(dataset.G), 0
            ids = state["perm"][state["pos"]:state["pos"] + args.batch]
            state["pos"] += args.batch
            return ids
        batches = [dataset.collate(sampler()) for _ in range(args.num_batches)]   # warm-up / per-launch timing leg
    elif args.scaling == "strong":
        # ONE global batch per step, the same on every rank (shared seed); each rank keeps the graphs partition_by_pairs gives it
        import numpy as np
        from kp_gnn_amd.dataset import KHopDataset
        if args.batch < world:
            raise SystemExit("--scaling strong needs --batch >= the number of ranks")
        shard_pairs = []
        for i in range(args.num_batches):
            host = make_batch(args, dp.shard_seed(0, args.num_batches, i, args.batch), threads)
            ds = KHopDataset.from_collated(host, host.node_ptr, device)
            shards = dp.partition_by_pairs(ds.h_pairs, world)
            b = ds.collate(np.asarray(shards[rank], dtype=np.int64))
            b._loss_weight = dp.shard_loss_weight(len(shards[rank]), args.batch)
            shard_pairs.append([int(sum(int(ds.h_pairs[g]) for g in sh)) for sh in shards])
            batches.append(b)
            del ds, host
    else:
        for i in range(args.num_batches):  # each rank owns its shard of graphs (distinct seeds)
            seed0 = dp.shard_seed(rank, args.num_batches, i, args.batch)
            b = make_batch(args, seed0, threads).to(device)
            b.build_csr()
            batches.append(b)
    torch.cuda.synchronize()
    t_data = time.perf_counter() - t_data
    if rank == 0:
        what = f"resident dataset of {args.dataset_graphs} graphs + " if args.fresh_batches else ""
        log(f"{what}{args.num_batches} batches x {args.batch} graphs built + CSR on device in {t_data:.1f}s "
            f"(N={batches[0].num_nodes}, E_khop={batches[0].csr.E}, A={batches[0].csr.A})")

    model = build_model(args, device)
    flat_grad = opt = None
    if args.train:
        if world > 1:  # identical replicas
            dp.broadcast_model(model)
        flat_grad = dp.flatten_grads(model)
        flat_param = dp.flatten_params(model)     # parameters and gradients: one flat bucket each (same order)
        flat_param.grad = flat_grad
        # elementwise: identical to per-parameter Adam (train_ZINC.py:244)
        opt = dp.FlatAdam(flat_param, flat_grad, lr=1e-3, device_step=(world == 1 and not args.no_graph and args.adam_in_graph))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        train_step(args, model, batches[i % len(batches)], opt, flat_grad, world)
    torch.cuda.synchronize()
    graphs = None
    static = static_graph = None
    if args.fresh_batches and not args.no_graph and args.train:
        # ONE hipGraph for every batch: capacity-shaped static buffers refilled in place by kpgnn_collate, the live node count on
        # the device (dataset.StaticBatch, ops.dynamic_rows).  The graph holds the collate launches, forward and backward.
        from kp_gnn_amd._lib import KpgnnError
        static = dataset.static_batch(args.batch)
        try:
            with static.dynamic():
                for _ in range(2):
                    static.stage(sampler())
                    static.launch_collate()
                    train_step(args, model, static.batch, opt, flat_grad, world)
                torch.cuda.synchronize()
                static.stage(sampler())
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    static.launch_collate()
                    fwd_bwd(args, model, static.batch, flat_grad)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                mode = "thread_local" if (torch.distributed.is_available() and torch.distributed.is_initialized()) else "global"
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode=mode):
                    static.launch_collate()
                    out_static = fwd_bwd(args, model, static.batch, flat_grad)
                static_graph = (g, out_static, False)
        except KpgnnError as e:        # a configuration without dynamic-row kernels: exact-shape batches, eager launches
            if rank == 0:
                log(f"static-shape graph not available for this configuration ({e}); --fresh-batches runs eagerly")
            static = static_graph = None
    if not args.no_graph and not args.fresh_batches:
        # (a capture failure is an error, not a silent downgrade to eager launches: --no-graph asks for those)
        graphs = capture_graphs(args, model, batches, flat_grad, opt if (args.train and world == 1 and args.adam_in_graph) else None)
        for i in range(len(batches)):  # one replayed step per graph before timing
            train_step(args, model, batches[i], opt, flat_grad, world, graphs[i])
        torch.cuda.synchronize()
    if rank == 0:
        log(f"{args.warmup} warm-up steps done; launch mode: {'hipGraph replay' if graphs else 'ONE static-shape hipGraph (collate + fwd + bwd)' if static_graph else 'eager'}")
    overflow_steps = 0
    if args.fresh_batches:
        from kp_gnn_amd.dataset import CapacityError
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if sampler is not None:        # a batch no step has seen: collated from the resident dataset, here, in the timed region
            ids = sampler()
            if static_graph is not None:
                try:
                    static.stage(ids)      # 16 KB of header to the device; the collate kernels are nodes of the graph
                    out_t = train_step(args, model, static.batch, opt, flat_grad, world, static_graph)
                    continue
                except CapacityError:      # (a batch beyond the static capacity: exact shapes, eager launches)
                    overflow_steps += 1
            out_t = train_step(args, model, dataset.collate(ids), opt, flat_grad, world)
            continue
        j = i % len(batches)
        out_t = train_step(args, model, batches[j], opt, flat_grad, world, graphs[j] if graphs else None)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = dp.max_over_ranks(elapsed, device, world)
    if args.scaling == "strong" and world > 1:      # the global mean loss = the sum of the ranks' weighted means
        out_t = out_t.clone()
        torch.distributed.all_reduce(out_t)
    final = float(out_t.float().abs().mean().item()) if not args.train else float(out_t.item())
    if rank == 0:
        log(f"timed {args.steps} steps in {elapsed:.3f}s")
    # Per-launch kernel timing (roofline leg): HIP events around every aggregation launch, on the launch
    # stream, over the same K steps run eagerly (events cannot bracket single kernels inside a replayed graph).
    timer = None
    if not args.no_roofline and rank == 0:
        timer = ops.LaunchTimer()
        ops.set_launch_timer(timer)
        train_step(args, model, batches[0], opt, flat_grad, 1)  # primes the byte-accounting caches
        timer.records.clear()
        for i in range(args.steps):
            train_step(args, model, batches[i % len(batches)], opt, flat_grad, 1)
        torch.cuda.synchronize()
        ops.set_launch_timer(None)

    if rank == 0:
        strong = args.scaling == "strong"
        total_graphs = args.batch * (1 if strong else world) * args.steps
        b0 = batches[0]
        what = "fwd+bwd+Adam, " + ("L1" if args.loss == "l1" else "MSE") + " loss" if args.train else "forward only (eval, no grad)"
        desc = {"zinc": "ZINC-12k-shaped synthetic molecules", "qm9": "QM9-shaped synthetic molecules",
                "regular": "random 3-regular graphs n=1280", "zinc_gd16": "ZINC-12k-shaped synthetic molecules"}[args.workload]
        out = {
            "metric": wl["metric"] if (args.model == wl["model"] or args.workload == "zinc") else f"graphs/sec {args.model} {args.workload}",
            "value": round(total_graphs / elapsed, 1),
            "unit": "graphs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32" if args.dtype == "f32" else "bf16 storage of the K-hop streams (hop-slot rows, S, dL/dS), f32 accumulate / parameters / states",
            "data": "synthetic",
            "config": {"workload": f"{desc}, {args.model} K={args.K} L={args.layers} h={args.hidden} kernel={args.kernel}"
                                   + (f" {args.combine} combine" if args.train else "") + f", {what}"
                                   + (", dense peripheral tensor" if args.dense_peripheral else ""),
                       "graphs_per_gpu_per_step": args.batch if not strong else round(args.batch / world, 1),
                       "global_batch": args.batch * (1 if strong else world),
                       "nodes_per_batch": b0.num_nodes, "khop_edges_per_batch": int(b0.csr.E),
                       "active_pairs_per_batch": int(b0.csr.A),
                       "dense_math": "fp32 operands and accumulation; products on the bf16 matrix cores from exact three-way bf16 "
                                     "splits (KPGNN_MATH_AUTO, bf3.h) where N >= 4096, else the fp32 matrix instruction",
                       "parallelism": f"dp{world}" + (" (one global batch partitioned by active pairs; rank 0's shard sizes are the "
                                                       "nodes / pairs fields; pairs per rank, first batch: " + str(shard_pairs[0]) + ")"
                                                       if strong else ""),
                       "launch": ("hipGraph replay of collate+fwd+bwd: ONE static-shape graph for all batches (capacity "
                                  f"{static.N_cap} nodes, live count on the device; {overflow_steps} eager overflow steps)") if static_graph
                       else ("hipGraph replay of fwd+bwd" if graphs else "eager"),
                       "batches": (f"fresh: every step collates a new shuffled subset of a resident {args.dataset_graphs}-graph dataset "
                                   "(kpgnn_collate inside the timed region)") if args.fresh_batches
                       else f"{args.num_batches} pre-staged batches cycled",
                       ("final_loss" if args.train else "mean_abs_output"): round(final, 5),
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
                out["roofline"]["measured_copy_GBps"] = round(stream_copy_gbps(device), 1)
            out["kernels"] = {k: {"launches": v["launches"], "avg_launch_ms": round(v["avg_ms"], 4),
                                  "algorithmic_GBps": round(v["gbps"], 1)} for k, v in s.items()}
            big = [r for r in timer.records if r[0] == "agg_fwd"]
            if big:   # the launch with the most algorithmic bytes (all K hops): its own roofline fraction
                top = max(r[1] for r in big)
                sel = [r for r in big if r[1] >= 0.999 * top]
                ms = sum(a.elapsed_time(b) for _, _, a, b in sel) / len(sel)
                out["kernels"]["agg_fwd_kmax"] = {"launches": len(sel), "avg_launch_ms": round(ms, 4),
                                                  "algorithmic_bytes_per_launch": int(top),
                                                  "algorithmic_GBps": round(top / (ms * 1e-3) / 1e9, 1),
                                                  "frac": round(top / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
            g = s.get("agg_bwd")
            if g:
                # (KP-GIN+ at N >= 4096: the pull form - agg_fwd_kernel over the (source, hop)-keyed CSR, ops.khop_pull_gather;
                #  elsewhere agg_bwd_kernel)
                pull = args.model == "KPGINPlus" and args.combine == "geometric" and args.dtype == "f32" and not args.dense_peripheral \
                    and b0.num_nodes >= 4096
                out["roofline_bwd"] = {"bound": "hbm", "kernel": "agg_fwd_kernel (pull form of the backward gather)" if pull else "agg_bwd_kernel",
                                       "achieved": round(g["gbps"], 1),
                                       "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(g["gbps"] / HBM_PEAK_GBPS, 4),
                                       "traffic": None,
                                       "launches": g["launches"], "avg_launch_ms": round(g["avg_ms"], 4),
                                       "algorithmic_bytes_per_launch": int(g["bytes_per_launch"])}
                out["roofline_bwd"].update(pmc_traffic(args, "agg_pull_kernel" if pull else "agg_bwd_kernel"))
        if world == 1 and not args.no_cpu_baseline:
            sd = {k: v for k, v in model.state_dict().items()}
            log(f"cpu baseline: oracle on {threads} host threads ...")
            out["cpu_baseline"] = cpu_baseline(args, sd, threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()        # rank 0 ran the per-launch timing leg alone: tear the communicator down together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
