"""Times the wide-output linear kernel (dx of the JK projection) against the BLAS library."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kp_gnn_amd import ops_dense
dev = torch.device("cuda:0")
N, I, O = 47450, 104, 936
dy = torch.randn(N, I, device=dev); w = torch.randn(I, O, device=dev) * 0.1
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1000
print("lib  dy@w  us", t(lambda: dy @ w))
for m in ("1", "2", "3"):
    os.environ["KPGNN_LIN_M"] = m
    print("wide m=%s us" % m, t(lambda: ops_dense._mfma_linear(dy, w, None, transposed=True)))
rep = torch.randn(N, O, device=dev)
print("lib dW us", t(lambda: dy.t() @ rep))
