"""Data parallelism for the hot path: one process per GPU, graphs sharded by rank, model replicated and
resident, ONE all-reduce of a flat fp32 gradient bucket per step (RCCL over xGMI on the GPU box; gloo in the
CPU tests).  Replaces the reference's single-process PyG DataParallel (train_ZINC.py:90-91,181-185), which
re-broadcasts the parameters and gathers outputs on GPU 0 every step.  The bucket is ~2 MB, i.e. latency
bound: one collective, no bucketing."""
import torch
import torch.distributed as dist


_ALIGN = 64   # floats: every tensor of a flat bucket starts on a 256-byte boundary (the kernels' 16-B vector paths)


def _offsets(params):
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
    return offs, off


def flatten_grads(model):
    """Make every parameter's .grad a view of one flat fp32 buffer; returns the buffer."""
    params = [p for p in model.parameters() if p.requires_grad]
    offs, total = _offsets(params)
    flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
    for p, off in zip(params, offs):
        p.grad = flat[off:off + p.numel()].view_as(p)
    return flat


def flatten_params(model):
    """Re-home every trainable parameter as a view of ONE flat fp32 buffer (same order as flatten_grads) and return
    it as an nn.Parameter: an elementwise optimiser (Adam, SGD) stepping that single tensor updates all of them with
    one multi-tensor launch instead of one chunk list per ~190 parameters; module attributes, state_dict keys and
    the captured graphs (which hold the views) are unaffected."""
    params = [p for p in model.parameters() if p.requires_grad]
    offs, total = _offsets(params)
    flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
    with torch.no_grad():
        for p, off in zip(params, offs):
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view_as(p)
    return torch.nn.Parameter(flat)


def grad_views(model):
    """(parameters, their .grad views into the flat bucket) in bucket order; flatten_grads must have run."""
    params = [p for p in model.parameters() if p.requires_grad]
    return params, [p.grad for p in params]


def copy_grads(views, grads):
    """views[i].copy_(grads[i]) for all i in one or two launches (kpgnn_multi_copy; the pointer table rides in the kernel
    arguments, so a captured graph replays it as is).  Pairs that are not contiguous fp32 CUDA tensors of equal size (a
    transposed weight gradient, say) go through the framework's multi-tensor copy."""
    import ctypes
    from . import _lib
    fast, slow = [], []
    for v, g in zip(views, grads):
        ok = (v.is_cuda and g.is_cuda and v.dtype == torch.float32 and g.dtype == torch.float32 and v.is_contiguous()
              and g.is_contiguous() and v.numel() == g.numel())
        (fast if ok else slow).append((v, g))
    if slow:
        torch._foreach_copy_([v for v, _ in slow], [g for _, g in slow])
    if not fast:
        return
    n = len(fast)
    src = (ctypes.c_void_p * n)(*[g.data_ptr() for _, g in fast])
    dst = (ctypes.c_void_p * n)(*[v.data_ptr() for v, _ in fast])
    cnt = (ctypes.c_int64 * n)(*[v.numel() for v, _ in fast])
    dev = fast[0][0].device
    with torch.cuda.device(dev):
        _lib.check(_lib.load().kpgnn_multi_copy(n, src, dst, cnt, torch.cuda.current_stream(dev).cuda_stream), "kpgnn_multi_copy")


class FlatAdam:
    """torch.optim.Adam (train_ZINC.py:244: lr, weight_decay = l2_wd) on the flat parameter bucket of flatten_params with the
    flat gradient bucket of flatten_grads: ONE elementwise launch per step (kpgnn_adam_step) instead of the framework's
    multi-tensor kernel over 65,536-element chunks (8 blocks, 42 us for 0.5 M parameters) plus its step-counter launch."""

    def __init__(self, flat_param, flat_grad, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, device_step=False):
        assert flat_param.is_cuda and flat_param.dtype == torch.float32 and flat_param.is_contiguous()
        assert flat_grad.shape == flat_param.shape and flat_grad.dtype == torch.float32 and flat_grad.is_contiguous()
        self.param, self.grad = flat_param, flat_grad
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat_param.data)
        self.exp_avg_sq = torch.zeros_like(flat_param.data)
        self.steps = 0
        # device_step: the step number lives in device memory (kpgnn_adam_step_device) - the launch can then be captured in
        # a hipGraph with the passes it follows; `steps` is not maintained on the host in that mode
        self.state = torch.zeros(2, dtype=torch.int64, device=flat_param.device) if device_step else None

    def zero_grad(self):
        self.grad.zero_()

    def step(self):
        from . import _lib
        from .ops_dense import invalidate_splits
        invalidate_splits()       # (the launch below changes the weights without bumping their version counters)
        dev = self.param.device
        if self.state is not None:
            with torch.cuda.device(dev):
                _lib.check(_lib.load().kpgnn_adam_step_device(
                    self.param.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                    self.param.numel(), self.state.data_ptr(), self.lr, self.betas[0], self.betas[1], self.eps,
                    self.weight_decay, torch.cuda.current_stream(dev).cuda_stream), "kpgnn_adam_step_device")
            return
        self.steps += 1
        with torch.cuda.device(dev):
            _lib.check(_lib.load().kpgnn_adam_step(
                self.param.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                self.param.numel(), self.steps, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                torch.cuda.current_stream(dev).cuda_stream), "kpgnn_adam_step")


def allreduce_mean(flat, world):
    """Mean of the flat gradient bucket over ranks.  EQUAL shards per rank only (bench.py's weak scaling): the mean of
    local-mean-loss gradients is then the gradient of the global mean loss the reference computes on GPU 0
    (train_ZINC.py:36,42).  Unequal shards: shard_loss_weight + allreduce_sum."""
    if world <= 1:
        return
    if flat.is_cuda and dist.get_backend() == "gloo":
        # rehearsal path only (gloo stages device tensors through the host): without this the collective's internal
        # stream waits fight the still-running graph replay of the ranks sharing one GPU (seconds per step)
        torch.cuda.synchronize(flat.device)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(world)


def partition_by_pairs(pairs_per_graph, world):
    """Shard the graphs of a global batch over `world` ranks balancing the ACTIVE (edge, hop) PAIRS, not the graph count:
    the aggregation's work and traffic follow A (ZINC-like graphs range over 9-37 nodes and E ~ n^2).
    Longest-processing-time greedy: graphs by descending weight, each to the currently lightest shard (ties: lowest rank).
    Returns `world` lists of graph indices, each ascending (a shard keeps the batch's node order).  Deterministic."""
    w = [int(v) for v in pairs_per_graph]
    order = sorted(range(len(w)), key=lambda i: (-w[i], i))
    shards, load = [[] for _ in range(world)], [0] * world
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        shards[r].append(i)
        load[r] += w[i]
    return [sorted(sh) for sh in shards]


def shard_loss_weight(local_graphs, global_graphs):
    """Factor for a rank's MEAN loss over its own graphs so that the SUM all-reduce of the gradients (allreduce_sum) is the
    gradient of the mean loss over the global batch, which is what the reference computes on GPU 0 after gathering the
    replicas' outputs (train_ZINC.py:34-36,42): sum_r (n_r / G) * mean_r == (1 / G) * sum_g loss_g."""
    return float(local_graphs) / float(global_graphs)


def allreduce_sum(flat, world):
    """Sum of the flat gradient bucket over ranks (unequal shards: every rank scaled its loss by shard_loss_weight)."""
    if world <= 1:
        return
    if flat.is_cuda and dist.get_backend() == "gloo":
        torch.cuda.synchronize(flat.device)     # (rehearsal path only, see allreduce_mean)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)


def select_graphs(node_ptr, edge_ptr, edge_index, edge_attr, x, idx):
    """Raw graphs (CSR-style concatenation, local node ids; the inputs of batch.collate_khop) restricted to the graphs
    `idx`, in that order: a rank's shard of a global batch."""
    import numpy as np
    nps, eps, eis, eas, xs = [0], [0], [], [], []
    for g in idx:
        n0, n1, e0, e1 = int(node_ptr[g]), int(node_ptr[g + 1]), int(edge_ptr[g]), int(edge_ptr[g + 1])
        nps.append(nps[-1] + n1 - n0)
        eps.append(eps[-1] + e1 - e0)
        eis.append(edge_index[:, e0:e1])
        if edge_attr is not None:
            eas.append(edge_attr[e0:e1])
        xs.append(x[n0:n1])
    cat = np.concatenate
    return (np.array(nps, dtype=np.int64), np.array(eps, dtype=np.int64),
            np.ascontiguousarray(cat(eis, axis=1)) if eis else np.zeros((2, 0), dtype=np.int64),
            np.ascontiguousarray(cat(eas)) if eas else None, np.ascontiguousarray(cat(xs)) if xs else x[:0])


def broadcast_model(model, src=0):
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src)


def shard_seed(rank, num_batches, batch_index, graphs_per_batch):
    """First graph seed of batch `batch_index` on `rank`: ranks and batches own disjoint seed ranges."""
    return (rank * num_batches + batch_index) * graphs_per_batch


def max_over_ranks(seconds, device, world):
    if world <= 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
