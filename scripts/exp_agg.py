"""GPU experiment: time the aggregation kernels in isolation on one ZINC-shaped batch (not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kp_gnn_amd import _lib, ops
from kp_gnn_amd.batch import synthetic_zinc_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
b = synthetic_zinc_batch(B, 0).to(dev)
csr = b.build_csr()
N, K, D = b.num_nodes, 8, 104
x = torch.randn(N, K, D, device=dev)
P = torch.randn(N, K, D, device=dev)
t0 = torch.randn(5, D, device=dev); tk = torch.randn(52, D, device=dev)
theta = torch.softmax(torch.randn(K, D, device=dev), 0)
g = torch.randn(N, K, D, device=dev)
print(f"N={N} A={csr.A} NKD tensor = {N*K*D*4/1e6:.1f} MB")

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

M = _lib.MODE_GINPLUS
for k in (8, 4, 1):
    xs, Ps, gs = x[:, :k], P[:, :k], g[:, :k].contiguous()
    th = theta[:k].contiguous()
    r = {}
    r["fwd tab+P+theta+pre"] = timeit(lambda: ops.aggregate_fwd_raw(csr, k, M, xs, t0, tk, Ps, None, th, None, True))
    r["fwd tab+P (out)"] = timeit(lambda: ops.aggregate_fwd_raw(csr, k, M, xs, t0, tk, Ps, None, None, None, False))
    r["fwd notab noP (sum)"] = timeit(lambda: ops.aggregate_fwd_raw(csr, k, _lib.MODE_SUM, xs, None, None, None, None, None, None, False))
    r["bwd tables"] = timeit(lambda: ops.aggregate_bwd_raw(csr, k, M, gs, None, 5, 52, True))
    r["bwd no tables"] = timeit(lambda: ops.aggregate_bwd_raw(csr, k, M, gs, None, 5, 52, False))
    r["table_grad"] = timeit(lambda: ops.table_grad_raw(csr, gs, 5, 52))
    cp = timeit(lambda: gs.clone())
    print(f"k={k}: " + "  ".join(f"{n}={v:.0f}us" for n, v in r.items()) + f"  | clone[N,k,D]={cp:.0f}us")
# table gather-sum
from kp_gnn_amd.body import _packed_peripheral_index
sizes = [5, 51] + [51] * 7
idx, off = _packed_peripheral_index(b.peripheral_edge_attr, b.peripheral_configuration_attr, sizes)[:2]
table = torch.randn(sum(sizes), D, device=dev); bias = torch.randn(D, device=dev)
go = torch.randn(N * K, D, device=dev)
print("tgs fwd us", timeit(lambda: ops.TableGatherSum.apply(table, bias, idx, off)))
tt = table.clone().requires_grad_(True)
def f():
    o = ops.TableGatherSum.apply(tt, None, idx, off); o.backward(go)
print("tgs fwd+bwd us", timeit(f))
u = torch.unique(idx, dim=0)
print("unique peripheral tuples:", u.shape[0], "of", idx.shape[0])
# new backward pieces
pre = torch.randn(N, K, D, device=dev); gh = torch.randn(N, D, device=dev)
uidx = torch.unique(idx.long(), dim=0, return_inverse=True)
uid = uidx[1].to(torch.int32).view(N, K).contiguous(); U = uidx[0].shape[0]
ptab = torch.randn(U, D, device=dev)
for k in (8, 4, 1):
    th = theta[:k].contiguous(); prek = pre[:, :k].contiguous(); uk = uid[:, :k]
    gk = g[:, :k].contiguous()
    a = timeit(lambda: ops.combine_bwd_raw(M, prek, gh, th, None, ptab, uk, True, False))
    b2 = timeit(lambda: ops.table_grad_raw(csr, gk, 5, 52, edges=True, uid=uk, n_dict=U, theta=th, gh=gh))
    c = timeit(lambda: ops.table_grad_raw(csr, gk, 5, 52, edges=True))
    f = timeit(lambda: ops.aggregate_fwd_raw(csr, k, M, x[:, :k], t0, tk, None, None, th, None, True, ptab=ptab, uid=uk))
    print(f"k={k}: combine_bwd={a:.0f}us table_grad(edges+dict)={b2:.0f}us table_grad(edges)={c:.0f}us fwd(dictP,theta,pre)={f:.0f}us")
