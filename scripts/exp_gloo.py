"""2-rank rehearsal micro-check: cost of the per-step collective on the gloo backend with device tensors."""
import os, time, torch, torch.distributed as dist
dist.init_process_group("gloo")
rank = dist.get_rank()
dev = torch.device("cuda:0")
t = torch.randn(530000, device=dev)
for n in (3, 10):
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        dist.all_reduce(t); t.div_(2)
    torch.cuda.synchronize()
    if rank == 0:
        print(f"gloo all_reduce of {t.numel()*4/1e6:.1f} MB device tensor: {(time.perf_counter()-t0)/n*1e3:.1f} ms", flush=True)
c = t.cpu()
t0 = time.perf_counter()
for _ in range(10):
    dist.all_reduce(c)
if rank == 0:
    print(f"gloo all_reduce of the same on CPU tensor: {(time.perf_counter()-t0)/10*1e3:.1f} ms", flush=True)
dist.destroy_process_group()
