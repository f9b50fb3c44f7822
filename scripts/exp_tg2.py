"""Ablation timing of kpgnn_table_grad (KPGNN_TG_DEBUG bits are read per launch by the host wrapper)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kp_gnn_amd import ops
from kp_gnn_amd.batch import synthetic_zinc_batch
dev = torch.device("cuda:0")
b = synthetic_zinc_batch(2048, 0).to(dev); csr = b.build_csr()
N, K = b.num_nodes, 8
print("N", N, "pairs", int(csr.tile_ptr[-1]), "tiles", csr.tile_ptr.numel() - 1)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
U = 25
uid = torch.randint(0, U, (N, K), dtype=torch.int32, device=dev)
uid_sorted = (torch.arange(N, device=dev).unsqueeze(1) // 23 % U).expand(N, K).to(torch.int32).contiguous()
for D in (13, 104):
    g = torch.randn(N, K, D, device=dev)
    for dbg in (0, 1, 2, 3, 4, 7):
        os.environ["KPGNN_TG_DEBUG"] = str(dbg)
        t1 = timeit(lambda: ops.table_grad_raw(csr, g, 5, 52, edges=True))
        t2 = timeit(lambda: ops.table_grad_raw(csr, g, 5, 52, edges=True, uid=uid, n_dict=U))
        print(f"D={D} dbg={dbg}: edges-only {t1:.0f} us   edges+dict(src2) {t2:.0f} us", flush=True)
