import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kp_gnn_amd import _lib, ops
from kp_gnn_amd.batch import synthetic_zinc_batch
dev = torch.device("cuda:0")
b = synthetic_zinc_batch(2048, 0).to(dev); csr = b.build_csr()
N, K, D = b.num_nodes, 8, 104
x = torch.randn(N, K, D, device=dev); P = torch.randn(N, K, D, device=dev)
t0 = torch.randn(5, D, device=dev); tk = torch.randn(52, D, device=dev)
theta = torch.softmax(torch.randn(K, D, device=dev), 0)
uid = torch.zeros(N, K, dtype=torch.int32, device=dev); ptab = torch.randn(25, D, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
G, S = _lib.MODE_GINPLUS, _lib.MODE_SUM
f = ops.aggregate_fwd_raw
r = {
 "sum: x->out": timeit(lambda: f(csr, K, S, x, None, None, None, None, None, None, False)),
 "sum+tab": timeit(lambda: f(csr, K, S, x, t0, tk, None, None, None, None, False)),
 "sum+tab+theta(hout only)": timeit(lambda: f(csr, K, S, x, t0, tk, None, None, theta, None, False)),
 "sum+tab+theta+pre": timeit(lambda: f(csr, K, S, x, t0, tk, None, None, theta, None, True)),
 "gelu+tab+theta+pre": timeit(lambda: f(csr, K, G, x, t0, tk, None, None, theta, None, True)),
 "gelu+tab+theta+pre+dictP": timeit(lambda: f(csr, K, G, x, t0, tk, None, None, theta, None, True, ptab=ptab, uid=uid)),
 "gelu+tab+theta+pre+denseP": timeit(lambda: f(csr, K, G, x, t0, tk, P, None, theta, None, True)),
}
for k, v in r.items(): print(f"{k:32s} {v:7.1f} us")
