"""A few launches of the table-gradient walk kernel at K = 8, D = 104 (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KPGNN_TG_KERNEL"] = "walk"
import torch
from kp_gnn_amd import ops
from kp_gnn_amd.batch import synthetic_zinc_batch
dev = torch.device("cuda:0")
b = synthetic_zinc_batch(2048, 0).to(dev); csr = b.build_csr()
N, K, D = b.num_nodes, 8, 104
g = torch.randn(N, K, D, device=dev)
for dbg in (0, 1):
    os.environ["KPGNN_TG_DEBUG"] = str(dbg)
    for _ in range(3):
        ops.table_grad_raw(csr, g, 5, 52, edges=True)
torch.cuda.synchronize()
