import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kp_gnn_amd import _lib, ops
from kp_gnn_amd.batch import synthetic_zinc_batch
dev = torch.device("cuda:0")
b = synthetic_zinc_batch(2048, 0).to(dev); csr = b.build_csr()
N, K, D = b.num_nodes, 8, 104
x = torch.randn(N, K, D, device=dev)
t0 = torch.randn(5, D, device=dev); tk = torch.randn(52, D, device=dev)
theta = torch.softmax(torch.randn(K, D, device=dev), 0)
for _ in range(3):
    ops.aggregate_fwd_raw(csr, K, _lib.MODE_SUM, x, t0, tk, None, None, theta, None, False)
torch.cuda.synchronize()
