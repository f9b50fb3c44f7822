"""Host enqueue time vs device time of an eager training step (where does `--no-graph` / `--fresh-batches` lose to graph replay?)."""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from kp_gnn_amd import dp
bench.dp = dp

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=2048)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--fresh", action="store_true")
a = ap.parse_args()
args = argparse.Namespace(workload="zinc", model="KPGINPlus", K=8, layers=8, hidden=104, batch=a.batch, kernel="spd", loss="l1",
                          train=True, combine="geometric", dtype="f32")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
model = bench.build_model(args, dev)
flat_grad = dp.flatten_grads(model)
flat_param = dp.flatten_params(model)
flat_param.grad = flat_grad
opt = dp.FlatAdam(flat_param, flat_grad, lr=1e-3, device_step=False)
if a.fresh:
    from kp_gnn_amd.dataset import KHopDataset
    saved, args.batch = args.batch, 10000
    host = bench.make_batch(args, 0, 16)
    args.batch = saved
    ds = KHopDataset.from_collated(host, host.node_ptr, dev)
    rng = np.random.default_rng(0)
    get = lambda i: ds.collate(rng.permutation(ds.G)[:args.batch])
else:
    bs = []
    for i in range(4):
        b = bench.make_batch(args, 1000 * i, 16).to(dev)
        b.build_csr()
        bs.append(b)
    get = lambda i: bs[i % 4]
for i in range(8):
    bench.train_step(args, model, get(i), opt, flat_grad, 1)
torch.cuda.synchronize()
st0 = torch.cuda.memory_stats()
t0 = time.perf_counter()
host = []
for i in range(a.steps):
    h0 = time.perf_counter()
    bench.train_step(args, model, get(i), opt, flat_grad, 1)
    host.append(time.perf_counter() - h0)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
st1 = torch.cuda.memory_stats()
print(f"batch {a.batch} fresh={a.fresh}: host enqueue {t_enq / a.steps * 1e3:.3f} ms/step, wall {t_all / a.steps * 1e3:.3f} ms/step; "
      f"per-step host min {min(host) * 1e3:.3f} max {max(host) * 1e3:.3f}")
for k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "allocation.all.allocated", "segment.all.allocated"):
    print("  ", k, st1.get(k, 0) - st0.get(k, 0))
print("   reserved MB", st1["reserved_bytes.all.current"] / 2**20, "allocated MB peak", st1["allocated_bytes.all.peak"] / 2**20)
# one sync'd step at a time: pure device time + launch latency
ts = []
for i in range(5):
    b = get(i)
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    bench.train_step(args, model, b, opt, flat_grad, 1)
    h1 = time.perf_counter()
    torch.cuda.synchronize()
    ts.append((h1 - h0, time.perf_counter() - h0))
print("   synced steps: host", [round(x[0] * 1e3, 3) for x in ts], "total", [round(x[1] * 1e3, 3) for x in ts])
