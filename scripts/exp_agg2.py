"""Ablation timing of the slot-mode forward gather (KPGNN_AGG_DEBUG bits are read per launch by the host wrapper)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kp_gnn_amd import _lib, ops
from kp_gnn_amd.batch import synthetic_zinc_batch
dev = torch.device("cuda:0")
b = synthetic_zinc_batch(2048, 0).to(dev)
csr = b.build_csr()
N, K, D = b.num_nodes, 8, 104
xs = [torch.randn(N, D, device=dev) for _ in range(K)]
t0 = torch.randn(5, D, device=dev); tk = torch.randn(52, D, device=dev)
theta = torch.softmax(torch.randn(K, D, device=dev), 0)
U = 25
uid = (torch.arange(N, device=dev).unsqueeze(1) * 7 + torch.arange(K, device=dev)).remainder(U).to(torch.int32).contiguous()
ptab = torch.randn(U, D, device=dev)
g = torch.randn(N, K, D, device=dev)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for dbg in (0, 1, 2, 4, 6, 7):
    os.environ["KPGNN_AGG_DEBUG"] = str(dbg)
    t = timeit(lambda: ops.aggregate_fwd_raw(csr, K, _lib.MODE_GINPLUS, None, t0, tk, None, None, theta, None, True, ptab=ptab, uid=uid, xs=xs))
    t2 = timeit(lambda: ops.aggregate_fwd_raw(csr, K, _lib.MODE_SUM, None, None, None, None, None, None, None, False, xs=xs))
    print(f"dbg={dbg}: fwd GIN+ fused {t:.0f} us   plain SUM (no tables/epilogue, [N,K,D] out) {t2:.0f} us", flush=True)
os.environ["KPGNN_AGG_DEBUG"] = "0"
print("bwd slots", round(timeit(lambda: ops.aggregate_bwd_raw(csr, K, _lib.MODE_GINPLUS, g, None, 5, 52, False, slots=True))), "us")
print("copy [N,K,D]", round(timeit(lambda: g.clone())), "us")
