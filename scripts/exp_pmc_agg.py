"""GPU experiment (run under rocprofv3 --pmc): a few launches of the slot-mode forward / backward gather on a
ZINC-shaped batch, exactly as the KP-GIN+ layer calls them (dictionary peripheral, fused geometric combine)."""
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
for _ in range(5):
    ops.aggregate_fwd_raw(csr, K, _lib.MODE_GINPLUS, None, t0, tk, None, None, theta, None, True, ptab=ptab, uid=uid, xs=xs)
    ops.aggregate_bwd_raw(csr, K, _lib.MODE_GINPLUS, g, None, 5, 52, False, slots=True)
torch.cuda.synchronize()
print("done")
