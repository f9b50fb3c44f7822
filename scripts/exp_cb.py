"""Isolated timing of kpgnn_combine_bwd (k = 8) for the activation modes: how much of it is erf arithmetic?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kp_gnn_amd import _lib, ops
dev = torch.device("cuda:0")
N, K, D = 47450, 8, 104
pre = torch.randn(N, K, D, device=dev); gh = torch.randn(N, D, device=dev)
theta = torch.softmax(torch.randn(K, D, device=dev), 0)
U = 25
uid = (torch.arange(N, device=dev).unsqueeze(1) * 7 + torch.arange(K, device=dev)).remainder(U).to(torch.int32).contiguous()
ptab = torch.randn(U, D, device=dev)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for name, mode in (("GINPLUS (erf)", _lib.MODE_GINPLUS), ("GCN (relu)", _lib.MODE_GCN), ("SUM (none)", _lib.MODE_SUM)):
    for gt in (True, False):
        t = timeit(lambda: ops.combine_bwd_raw(mode, pre, gh, theta, None, ptab, uid, want_gtheta=gt, want_gv=False))
        print(f"{name:14s} gtheta={gt}: {t:.0f} us", flush=True)
os.environ["KPGNN_CB_DEBUG"] = "1"
print("GINPLUS gtheta, no uid/dictionary read:", round(timeit(lambda: ops.combine_bwd_raw(_lib.MODE_GINPLUS, pre, gh, theta, None, ptab, uid, want_gtheta=True, want_gv=False))), "us")
for dbg in (2, 4, 6, 7):
    os.environ["KPGNN_CB_DEBUG"] = str(dbg)
    print(f"GINPLUS gtheta dbg={dbg}:", round(timeit(lambda: ops.combine_bwd_raw(_lib.MODE_GINPLUS, pre, gh, theta, None, ptab, uid, want_gtheta=True, want_gv=False))), "us")
os.environ["KPGNN_CB_DEBUG"] = "0"
print("GINPLUS gtheta, dense P:", round(timeit(lambda: ops.combine_bwd_raw(_lib.MODE_GINPLUS, pre, gh, theta, pre, None, None, want_gtheta=True, want_gv=False))), "us")
print("clone [N,K,D]:", round(timeit(lambda: pre.clone())), "us")
