"""Run-to-run bitwise reproducibility of the bf16-split kernels (JK forward, its dX, the weight-gradient pair)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kp_gnn_amd.ops_dense import JKConcatLinear
dev = torch.device("cuda:0")
for (N, H, S, O) in ((6917, 32, 5, 32), (47450, 104, 9, 104), (4099, 104, 9, 104)):
    g = torch.Generator().manual_seed(1)
    states = [torch.randn(N, H, generator=g).to(dev) for _ in range(S)]
    w = (torch.randn(O, S * H, generator=g) * 0.05).to(dev)
    b = torch.randn(O, generator=g).to(dev)
    gy = torch.randn(N, O, generator=g).to(dev)
    outs = []
    for rep in range(4):
        junk = torch.full((rep * 1000003 + 17,), float(rep), device=dev)        # shifts the allocator's choices
        sd = [t.clone().requires_grad_(True) for t in states]
        wd, bd = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = JKConcatLinear.apply(wd, bd, *sd)
        (y * gy).sum().backward()
        outs.append((y.detach().clone(), wd.grad.clone(), bd.grad.clone(), [t.grad.clone() for t in sd]))
        del junk
    ok_y = all(torch.equal(outs[0][0], o[0]) for o in outs)
    ok_w = all(torch.equal(outs[0][1], o[1]) for o in outs)
    ok_b = all(torch.equal(outs[0][2], o[2]) for o in outs)
    ok_x = all(all(torch.equal(a, c) for a, c in zip(outs[0][3], o[3])) for o in outs)
    print(f"N={N} H={H} S={S} O={O}: y {ok_y}  dW {ok_w}  db {ok_b}  dX {ok_x}")
