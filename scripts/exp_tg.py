"""Time kpgnn_table_grad on the launches of one real bench step (default workload):
    python scripts/exp_tg.py [variant ...]       variant = integer `kernel` selector passed through (0 = product)
Captures the arguments of every table_grad call of one forward+backward, then replays each call alone."""
import argparse
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench                                    # noqa: E402
from kp_gnn_amd import ops                      # noqa: E402


def main():
    variants = [int(v, 0) for v in sys.argv[1:]] or [0]
    wl = bench.WORKLOADS["zinc"]
    args = argparse.Namespace(workload="zinc", combine="geometric", **{k: wl[k] for k in ("model", "K", "layers", "hidden", "batch", "kernel", "loss", "train")})
    dev = torch.device("cuda:0")
    model = bench.build_model(args, dev)
    batch = bench.make_batch(args, 0, 8).to(dev)
    batch.build_csr()
    calls = []
    real = ops.table_grad_raw

    def spy(csr, g, n0, nk, **kw):
        calls.append((csr, g.clone(), n0, nk, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()}))
        return real(csr, g, n0, nk, **kw)

    ops.table_grad_raw = spy
    loss = bench.loss_of(args, model(batch), batch.y)
    loss.backward()
    ops.table_grad_raw = real
    torch.cuda.synchronize()
    print("captured", len(calls), "calls")
    for ci, (csr, g, n0, nk, kw) in enumerate(calls):
        kw = dict(kw)
        kw.pop("kernel", None)
        desc = f"call {ci}: g {tuple(g.shape)} n0 {n0} nk {nk} edges {kw.get('edges', True)} U {kw.get('n_dict', 0)}"
        ref = None
        out = []
        for v in variants:
            for _ in range(3):
                res = real(csr, g, n0, nk, kernel=v, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                res = real(csr, g, n0, nk, kernel=v, **kw)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            if ref is None:
                ref = res
                err = 0.0
            else:
                err = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(res, ref) if a is not None)
            out.append(f"v{v:#x} {us:6.1f}us err {err:.1e}")
        print(desc, " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
