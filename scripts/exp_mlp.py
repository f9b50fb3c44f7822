"""Experiment driver (GPU box): fused MLP backward error localisation + weight-gradient launch timings."""
import copy, ctypes, sys, time
import torch
sys.path.insert(0, ".")
from kp_gnn_amd import _lib, ops_dense
from kp_gnn_amd.ops_dense import mlp_linear_bn_relu_x2

dev = torch.device("cuda:0")
lib = _lib.load()


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for N in (20000, 32768, 32769, 47450):
    I = O = 104
    g = torch.Generator().manual_seed(N)
    ref = torch.nn.Sequential(torch.nn.Linear(I, O), torch.nn.BatchNorm1d(O), torch.nn.ReLU(),
                              torch.nn.Linear(O, O), torch.nn.BatchNorm1d(O), torch.nn.ReLU())
    hip = copy.deepcopy(ref).to(dev).train()
    x = torch.randn(N, I, generator=g)
    w = torch.randn(N, O, generator=g)
    xr = x.clone().requires_grad_(True)
    (ref(xr) * w).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    out = mlp_linear_bn_relu_x2(hip, xd)
    (out * w.to(dev)).sum().backward()
    err = (xd.grad.cpu() - xr.grad).abs()
    rows = err.max(dim=1).values
    bad = torch.nonzero(rows > 1e-3).flatten()
    print(f"N={N}: max dx err {float(err.max()):.3e}; bad rows {bad.numel()} first {bad[:5].tolist()} last {bad[-5:].tolist()}",
          "dW3 err %.2e" % float((hip[3].weight.grad.cpu() - ref[3].weight.grad).abs().max()),
          "dW0 err %.2e" % float((hip[0].weight.grad.cpu() - ref[0].weight.grad).abs().max()), flush=True)

# ---- weight gradient timings: two single launches vs the pair
N, O, I = 47450, 104, 104
dy1, dy2, x1, x2 = (torch.randn(N, 104, device=dev) for _ in range(4))
m, i, gg, b = (torch.rand(104, device=dev) + 0.5 for _ in range(4))
dw = torch.empty(2, O, I, device=dev); db = torch.empty(2, O, device=dev)
nb = 2 * int(lib.kpgnn_wgrad_workspace_bytes(O, I))
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream


def desc(dy, x, k, tr):
    q = _lib.WgradDesc()
    q.N, q.O, q.I = N, O, I
    q.dy, q.dy_stride, q.x, q.x_stride = dy.data_ptr(), O, x.data_ptr(), I
    q.dw, q.db = dw[k].data_ptr(), db[k].data_ptr()
    q.workspace, q.workspace_bytes = ws.data_ptr(), nb
    if tr:
        q.x_mean, q.x_invstd, q.x_gamma, q.x_beta, q.x_relu = m.data_ptr(), i.data_ptr(), gg.data_ptr(), b.data_ptr(), 1
    return q


a0, b0 = desc(dy1, x1, 0, False), desc(dy2, x2, 1, False)
a1 = desc(dy1, x1, 0, True)
print("single, plain      %.1f us" % timeit(lambda: lib.kpgnn_linear_wgrad(ctypes.byref(a0), st)))
print("single, transform  %.1f us" % timeit(lambda: lib.kpgnn_linear_wgrad(ctypes.byref(a1), st)))
print("pair, plain        %.1f us" % timeit(lambda: lib.kpgnn_linear_wgrad_pair(ctypes.byref(a0), ctypes.byref(b0), st)))
print("pair, transform    %.1f us" % timeit(lambda: lib.kpgnn_linear_wgrad_pair(ctypes.byref(a1), ctypes.byref(b0), st)))
