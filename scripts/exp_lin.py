"""Isolated timing of kpgnn_linear_fwd ([N,104] x [104,104]) with ablation bits (KPGNN_LIN_DEBUG: 1 no MFMA loop, 2 no output)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kp_gnn_amd import ops_dense
ops_dense._USE_MFMA_LINEAR = True
dev = torch.device("cuda:0")
N, D = 47450, 104
x = torch.randn(N, D, device=dev); w = torch.randn(D, D, device=dev) * 0.1; b = torch.randn(D, device=dev)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for mm, gg in (("1", "256"), ("1", "512"), ("2", "256"), ("1", "768"), ("2", "384"), ("3", "256")):
    os.environ["KPGNN_LIN_M"] = mm; os.environ["KPGNN_LIN_GRID"] = gg
    print("M =", mm, "grid", gg, round(timeit(lambda: ops_dense._mfma_linear(x, w, b)), 1), "us", flush=True)
os.environ.pop("KPGNN_LIN_M"); os.environ.pop("KPGNN_LIN_GRID")
for dbg in (0, 1, 2, 3, 4):
    os.environ["KPGNN_LIN_DEBUG"] = str(dbg)
    print(f"dbg={dbg}: mfma linear {timeit(lambda: ops_dense._mfma_linear(x, w, b)):.1f} us", flush=True)
os.environ["KPGNN_LIN_DEBUG"] = "0"
print("torch F.linear", round(timeit(lambda: torch.nn.functional.linear(x, w, b)), 1), "us")
print("torch x @ w   ", round(timeit(lambda: x @ w), 1), "us")
