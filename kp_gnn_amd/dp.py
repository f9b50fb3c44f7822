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


def allreduce_mean(flat, world):
    """Mean of the flat gradient bucket over ranks.  Equal shards per rank => the mean of local-mean-loss
    gradients is the gradient of the global mean loss the reference computes on GPU 0 (train_ZINC.py:36,42)."""
    if world <= 1:
        return
    if flat.is_cuda and dist.get_backend() == "gloo":
        # rehearsal path only (gloo stages device tensors through the host): without this the collective's internal
        # stream waits fight the still-running graph replay of the ranks sharing one GPU (seconds per step)
        torch.cuda.synchronize(flat.device)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(world)


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
