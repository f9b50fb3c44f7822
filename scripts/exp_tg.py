import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kp_gnn_amd import ops
from kp_gnn_amd.batch import synthetic_zinc_batch
dev = torch.device("cuda:0")
b = synthetic_zinc_batch(2048, 0).to(dev); csr = b.build_csr()
N, K, D = b.num_nodes, 8, 104
g = torch.randn(N, K, D, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for k in (8, 1):
    gk = g[:, :k].contiguous()
    print("k", k, "table_grad us", round(timeit(lambda: ops.table_grad_raw(csr, gk, 5, 52, edges=True))))
