"""HIP-event time of ONE kpgnn_dict_grad_multi launch at the bench shape (8 layers, k = 1..8, N = 47,450, D = 104) on
molecule-like ids (one id covers most nodes of a hop), gh cold (a 512-MB fill between launches), against 8 kpgnn_dict_grad."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kp_gnn_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
N, K, D, U, L = 47450, 8, 104, int(os.environ.get("UB_U", 25)), 8
frac = [0.999, 0.95, 0.9, 0.86, 0.84, 0.83, 0.82, 0.82]
uid = torch.zeros((N, K), dtype=torch.int32)
for k in range(K):
    other = torch.rand(N) > frac[k]
    uid[other, k] = torch.randint(1, U, (int(other.sum()),), dtype=torch.int32)
uid = uid.to(dev)
dom = torch.zeros(K, dtype=torch.int32, device=dev)
ghs = [torch.randn(N, D, device=dev) for _ in range(L)]
thetas = [torch.rand(l + 1, D, device=dev) for l in range(L)]
items = []
for l in range(L):
    u = uid[:, :l + 1]
    u._kp_dom = dom[:l + 1].contiguous()
    items.append((u, thetas[l], ghs[l]))
flush = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev)


def timed(fn, reps=5):
    ts = []
    for _ in range(reps):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts), out


t1, g1 = timed(lambda: ops.dict_grad_multi_raw(items, U))
t0, g0 = timed(lambda: sum(ops.dict_grad_raw(u, U, th, gh) for u, th, gh in items))
print(f"U={U}: one launch for {L} layers {t1:.1f} us (incl. its slab reduce); {L} launches {t0:.1f} us; max |diff| {float((g1 - g0).abs().max()):.3e} of {float(g0.abs().max()):.3e}")
