"""Data parallelism for the hot path: one process per GPU, graphs sharded by rank, model replicated and
resident, ONE all-reduce of a flat fp32 gradient bucket per step (RCCL over xGMI on the GPU box; gloo in the
CPU tests).  Replaces the reference's single-process PyG DataParallel (train_ZINC.py:90-91,181-185), which
re-broadcasts the parameters and gathers outputs on GPU 0 every step.  The bucket is ~2 MB, i.e. latency
bound: one collective, no bucketing."""
import torch
import torch.distributed as dist


def flatten_grads(model):
    """Make every parameter's .grad a view of one flat fp32 buffer; returns the buffer."""
    params = [p for p in model.parameters() if p.requires_grad]
    flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=params[0].device)
    off = 0
    for p in params:
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    return flat


def grad_views(model):
    """(parameters, their .grad views into the flat bucket) in bucket order; flatten_grads must have run."""
    params = [p for p in model.parameters() if p.requires_grad]
    return params, [p.grad for p in params]


def allreduce_mean(flat, world):
    """Mean of the flat gradient bucket over ranks.  Equal shards per rank => the mean of local-mean-loss
    gradients is the gradient of the global mean loss the reference computes on GPU 0 (train_ZINC.py:36,42)."""
    if world <= 1:
        return
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
