"""Per-launch timing of the dense-tail kernels at the bench shape (N = 47,450, H = 104): HIP events over repeated launches."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kp_gnn_amd import _lib

lib = _lib.load()
_lib.DENSE_MATH = {"auto": _lib.MATH_AUTO, "f32": _lib.MATH_F32}.get(os.environ.get("UB_MATH", "auto"), None)
if _lib.DENSE_MATH is None:
    _lib.DENSE_MATH = int(os.environ["UB_MATH"])   # UB_MATH=f32: fp32 matrix instruction
print("dense math:", os.environ.get("UB_MATH", "auto"))
dev = torch.device("cuda:0")
N, H, S = int(os.environ.get("UB_N", 47450)), 104, 9
st = torch.cuda.current_stream().cuda_stream


def timeit(name, fn, flops=None, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    extra = f"  {flops / us / 1e6:7.1f} TFLOP/s (useful)" if flops else ""
    print(f"{name:42s} {us:8.1f} us{extra}", flush=True)


x = torch.randn(N, H, device=dev)
dy = torch.randn(N, H, device=dev)
w = torch.randn(H, H, device=dev) * 0.1
b = torch.randn(H, device=dev)
y = torch.empty(N, H, device=dev)

# plain linear (lin_fused PRO 0 EPI 0)
d = _lib.LinearDesc()
d.N, d.O, d.I = N, H, H
d.x, d.x_stride, d.w, d.bias, d.y, d.y_stride = x.data_ptr(), H, w.data_ptr(), b.data_ptr(), y.data_ptr(), H
timeit("linear_fwd [N,104]x[104,104]", lambda: _lib.check(lib.kpgnn_linear_fwd(ctypes.byref(d), st), "lin"), 2.0 * N * H * H)

# wgrad single / pair
dw = torch.empty(2, H, H, device=dev)
db = torch.empty(2, H, device=dev)
nb = 2 * int(lib.kpgnn_wgrad_workspace_bytes(H, H))
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
x2 = torch.randn(N, H, device=dev)
dy2 = torch.randn(N, H, device=dev)
qa, qb = _lib.WgradDesc(), _lib.WgradDesc()
for q, a_dy, a_x, k in ((qa, dy, x, 0), (qb, dy2, x2, 1)):
    q.N, q.O, q.I = N, H, H
    q.dy, q.dy_stride, q.x, q.x_stride = a_dy.data_ptr(), H, a_x.data_ptr(), H
    q.dw, q.db = dw[k].data_ptr(), db[k].data_ptr()
    q.workspace, q.workspace_bytes = ws.data_ptr(), nb
timeit("wgrad single (+reduce)", lambda: _lib.check(lib.kpgnn_linear_wgrad(ctypes.byref(qa), st), "wg"), 2.0 * N * H * H)
timeit("wgrad pair (+reduce)", lambda: _lib.check(lib.kpgnn_linear_wgrad_pair(ctypes.byref(qa), ctypes.byref(qb), st), "wgp"), 4.0 * N * H * H)
job = _lib.ReduceJob()
qa.defer = ctypes.cast(ctypes.pointer(job), ctypes.c_void_p)
timeit("wgrad pair (reduce deferred)", lambda: _lib.check(lib.kpgnn_linear_wgrad_pair(ctypes.byref(qa), ctypes.byref(qb), st), "wgp"), 4.0 * N * H * H)
qa.defer = None

ref = dy.double().t() @ x.double()
err = ((dw[0].double() - ref).abs().max() / ref.abs().max()).item()
print(f"wgrad dW max error / max |dW| against float64: {err:.3e};  db error {((db[0].double() - dy.double().sum(0)).abs().max() / dy.double().sum(0).abs().max()).item():.3e}")

# JK: grouped forward, blocked dX, grouped wgrad
states = [torch.randn(N, H, device=dev) for _ in range(S)]
wj = torch.randn(H, S * H, device=dev) * 0.05
yj = torch.empty(N, H, device=dev)
g = _lib.LinearGroupDesc()
g.N, g.O, g.I, g.group = N, H, H, S
for l, t in enumerate(states):
    g.x[l] = t.data_ptr()
g.x_stride, g.w, g.bias, g.y, g.relu = H, wj.data_ptr(), b.data_ptr(), yj.data_ptr(), 1
wsl = torch.empty(int(lib.kpgnn_linear_split_workspace_bytes(H, H, S)), dtype=torch.uint8, device=dev)
g.workspace, g.workspace_bytes = wsl.data_ptr(), wsl.numel()
timeit("JK linear_group_fwd (9 states)", lambda: _lib.check(lib.kpgnn_linear_group_fwd(ctypes.byref(g), st), "lg"), 2.0 * N * H * H * S)
G = torch.empty(S, N, H, device=dev)
dl = _lib.LinearDesc()
dl.N, dl.O, dl.I = N, S * H, H
dl.x, dl.x_stride, dl.w, dl.y, dl.y_stride = dy.data_ptr(), H, wj.data_ptr(), G.data_ptr(), H
dl.w_transposed, dl.y_block_cols, dl.y_block_stride, dl.x_mask = 1, H, N * H, yj.data_ptr()
wsd = torch.empty_like(wsl)
dl.workspace, dl.workspace_bytes = wsd.data_ptr(), wsd.numel()
timeit("JK dX linear_wide (masked)", lambda: _lib.check(lib.kpgnn_linear_fwd(ctypes.byref(dl), st), "lw"), 2.0 * N * H * H * S)
refy = torch.relu(torch.cat(states, 1).double() @ wj.double().t() + b.double())
print(f"JK forward max error / max |y| against float64: {((yj.double() - refy).abs().max() / refy.abs().max()).item():.3e}")
refg = (dy.double() * (yj > 0)) @ wj.double()
print(f"JK dX max error / max |dX| against float64: {((G.permute(1, 0, 2).reshape(N, S * H).double() - refg).abs().max() / refg.abs().max()).item():.3e}")
dwj = torch.empty(H, S * H, device=dev)
dbj = torch.empty(H, device=dev)
nbj = int(lib.kpgnn_wgrad_group_workspace_bytes(H, H, S))
wsj = torch.empty(nbj, dtype=torch.uint8, device=dev)
qj = _lib.WgradDesc()
qj.N, qj.O, qj.I = N, H, H
qj.dy, qj.dy_stride, qj.x, qj.x_stride, qj.dy_mask = dy.data_ptr(), H, states[0].data_ptr(), H, yj.data_ptr()
qj.dw, qj.db, qj.workspace, qj.workspace_bytes = dwj.data_ptr(), dbj.data_ptr(), wsj.data_ptr(), nbj
xs = (ctypes.c_void_p * S)(*[t.data_ptr() for t in states])
timeit("JK wgrad_group (9 problems, +reduce)", lambda: _lib.check(lib.kpgnn_linear_wgrad_group(ctypes.byref(qj), xs, S, st), "wgg"), 2.0 * N * H * H * S)
# library references
rep = torch.cat(states, dim=1)
timeit("torch cat(9 states)", lambda: torch.cat(states, dim=1))
timeit("torch addmm [N,936]x[936,104]", lambda: torch.addmm(b, rep, wj.t()), 2.0 * N * H * H * S)
timeit("torch dW = dy^T rep", lambda: dy.t() @ rep, 2.0 * N * H * H * S)
timeit("torch x @ w.t() [N,104]x[104,104]", lambda: x @ w.t(), 2.0 * N * H * H)
timeit("device copy [N,104] fp32", lambda: y.copy_(x))
