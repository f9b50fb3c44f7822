"""Autograd operators over the C ABI (include/kpgnn.h).  Host-side plumbing only: tensors in, pointers
and strides out; every arithmetic step of the K-hop aggregation runs in the HIP kernels."""
import ctypes

import torch

from . import _lib
from ._lib import MODE_GCN, MODE_GIN, MODE_GINPLUS, MODE_SUM

_SQRT1_2 = 0.7071067811865476
_INV_SQRT_2PI = 0.3989422804014327


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


class LaunchTimer:
    """Opt-in per-launch timing of the aggregation kernels with HIP events recorded on the stream the
    kernel is launched on (bench.py's roofline leg).  Each record: (kind, algorithmic_bytes, start, stop)."""

    def __init__(self):
        self.records = []

    def summary(self):
        """kind -> dict(launches, avg_ms, bytes_per_launch, gbps); call after a device synchronize."""
        out = {}
        for kind, nbytes, a, b in self.records:
            d = out.setdefault(kind, {"launches": 0, "ms": 0.0, "bytes": 0})
            d["launches"] += 1
            d["ms"] += a.elapsed_time(b)
            d["bytes"] += nbytes
        for d in out.values():
            d["avg_ms"] = d["ms"] / d["launches"]
            d["bytes_per_launch"] = d["bytes"] / d["launches"]
            d["gbps"] = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
        return out


_timer = None

def set_launch_timer(timer):
    global _timer
    _timer = timer


def algorithmic_bytes(csr, k_act, D, n_tensors, n_rows_tables, extra_nd=0, s=4):
    """s*N*k*D per streamed [N,k,D] tensor + (4+2) B per active pair + int32 row pointers
    + the tables once (+ 4*N*D per fp32 [N,D] tensor), s = 4 (fp32) or 2 (bf16 storage).  n_tensors may be fractional
    when the streams of one launch have different widths (it counts fp32-equivalents at s = 4)."""
    A = csr.active_pairs(k_act)
    return int(s * csr.N * k_act * D * n_tensors) + A * 6 + 4 * (csr.N * csr.K + 1) + 4 * D * n_rows_tables \
        + 4 * csr.N * D * extra_nd


# Storage of the big per-(node,hop) streams of the fused KP-GIN+ path (hop-slot rows, S saved for the backward, dL/dS):
# torch.float32 (default) or torch.bfloat16 (kpgnn.h KPGNN_STORE_BF16: 2-byte rows, fp32 sums).  Other paths ignore it.
_STORAGE = torch.float32


def set_storage_dtype(dtype):
    """torch.float32 / torch.bfloat16 (or "fp32" / "bf16"): see _STORAGE.  Returns the previous setting."""
    global _STORAGE
    prev = _STORAGE
    dtype = {"fp32": torch.float32, "f32": torch.float32, "bf16": torch.bfloat16}.get(dtype, dtype)
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError(f"storage dtype must be float32 or bfloat16, got {dtype}")
    _STORAGE = dtype
    return prev


def bf16_shadow(t):
    """bf16 copy of a hop state [N,D] for the gathers of later layers (made once per state, kept on the tensor)."""
    rec = getattr(t, "_kp_bf16", None)
    if rec is None or rec[0] != t._version:      # (a state modified in place after its first reader gets a fresh shadow)
        rec = (t._version, t.detach().to(torch.bfloat16).contiguous())
        try:
            t._kp_bf16 = rec
        except AttributeError:
            pass
    return rec[1]


def _ptr(t):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------ dynamic row count (static-shape graphs)
# A hipGraph bakes every kernel argument in, the row count N of a batch included, and no two shuffled batches have the same
# N.  Under `dynamic_rows(count, capacity)` every launch that works on `capacity` rows is handed `count` (a device int32[1],
# e.g. the node count kpgnn_collate leaves in its header) as its descriptor's n_dyn: tensors are allocated for the capacity,
# kernels read, write and SUM only the live rows - so ONE captured graph serves every batch up to the capacity
# (dataset.StaticBatch, bench.py --fresh-batches).  Operators without an n_dyn field refuse to run in this mode.
_DYN = None     # (count tensor, capacity)


class dynamic_rows:
    def __init__(self, count, capacity):
        assert count.dtype == torch.int32 and count.numel() >= 1 and count.is_cuda
        self._new = (count, int(capacity))

    def __enter__(self):
        global _DYN
        self._old, _DYN = _DYN, self._new
        return self

    def __exit__(self, *exc):
        global _DYN
        _DYN = self._old
        return False


def dyn_ptr(rows):
    """n_dyn for a launch over `rows` rows: the live-count pointer when `rows` is the capacity of the active dynamic_rows
    block, else None (launches over other row counts - graphs, dictionary rows - are static)."""
    if _DYN is not None and int(rows) == _DYN[1]:
        return _DYN[0].data_ptr()
    return None


def refuse_dynamic_rows(what, rows):
    if _DYN is not None and int(rows) == _DYN[1]:
        raise _lib.KpgnnError(f"{what} has no dynamic row count (n_dyn): run this configuration on exact-shape batches")


def _last_contig(t):
    """[N,K,D] view with unit innermost stride (copy only if the caller handed something exotic)."""
    return t if t.stride(-1) == 1 else t.contiguous()


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.KpgnnError("kp_gnn_amd ops need CUDA/HIP tensors: there is no CPU fallback "
                                  "(the CPU restatement lives in oracle/ and is test-only)")


def _check_codes(csr, k_act, table0, tablek):
    if csr.A == 0:
        return
    if csr.max_code0 >= table0.shape[0]:
        raise IndexError(f"edge code {csr.max_code0} out of range for hop1_edge_emb with {table0.shape[0]} rows")
    if k_act > 1 and tablek is not None and csr.max_codek >= tablek.shape[0]:
        raise IndexError(f"edge code {csr.max_codek} out of range for hopk_edge_emb with {tablek.shape[0]} rows")


# ------------------------------------------------------------------------------------ deferred reductions
# A weight-gradient launch leaves per-block partial sums that a small second launch adds up.  Inside
# `deferred_reductions()` that second launch is not issued: the job (kpgnn_reduce_job) waits in this list and the next
# table-gradient call - which has a finishing launch of its own - takes it along (8 launches less per step at L = 8); what
# is left when the block ends is run then.  The gradient tensors such a backward returns are NOT valid until the block
# ends: only for callers that read gradients afterwards (torch.autograd.grad(...) inside the block, or .backward() into
# parameters whose .grad is None) - never with gradient accumulation into existing .grad tensors or backward hooks.
_pending_reduce = None        # None: off; list of (ReduceJob, tensors kept alive)


class deferred_reductions:
    def __enter__(self):
        global _pending_reduce
        self._outer = _pending_reduce
        if _pending_reduce is None:
            _pending_reduce = []
        return self

    def __exit__(self, *exc):
        global _pending_reduce
        if self._outer is None:
            jobs, _pending_reduce = _pending_reduce, None
            _deferred_owners.clear()
            if jobs and exc[0] is None:
                flush_reductions(jobs)
        return False


def flush_reductions(jobs):
    arr = (_lib.ReduceJob * len(jobs))(*[j for j, _ in jobs])
    dev = jobs[0][1][0].device
    with torch.cuda.device(dev):
        _lib.check(_lib.load().kpgnn_reduce_jobs(arr, len(jobs), torch.cuda.current_stream(dev).cuda_stream), "kpgnn_reduce_jobs")


_deferred_owners = set()      # data_ptr of the parameters whose gradient a queued job will write


def defer_reduce_job(*params):
    """A fresh job slot when reductions are being deferred (the caller hands ctypes.byref(job) to a launch with a `defer`
    field, then calls queue_reduce_job), else None.
    `params`: the parameters whose gradients the job will write.  A parameter that feeds TWO nodes (shared MLP weights, a
    layer applied twice) has its two gradients summed by autograd as soon as both nodes have returned - before the block
    ends - so a second job for a parameter that already has one is refused (None: the caller reduces at once) and everything
    queued so far is run first: the sum then only ever reads finished gradients."""
    if _pending_reduce is None:
        return None
    keys = {p.data_ptr() for p in params if p is not None}
    if keys & _deferred_owners:
        jobs = list(_pending_reduce)
        del _pending_reduce[:]
        _deferred_owners.clear()
        if jobs:
            flush_reductions(jobs)
        return None
    _deferred_owners.update(keys)
    return _lib.ReduceJob()


def queue_reduce_job(job, keep_alive):
    _pending_reduce.append((job, keep_alive))


def take_reduce_job():
    """The oldest waiting job, or None: for a call whose finishing launch can take one along."""
    if _pending_reduce:
        return _pending_reduce.pop(0)
    return None


class DictPeripheral:
    """Dictionary form of the peripheral features: P[i,k,:] = table[uid[i,k], :].

    The (node,hop) peripheral-subgraph feature tuples repeat massively (25 distinct among 379,600 slots of a
    2048-molecule batch), so instead of a dense [N,K,D] tensor the callers may hand the layers this object
    as `peripheral_attr`: `table` [U,D] carries the gradient, `uid` [N,K] (int32) is static per batch.
    Supports the body's `peripheral_attr[:, :k]` hop-prefix slicing (models/GNNs.py:421-423)."""

    def __init__(self, table, uid):
        self.table, self.uid = table, uid

    @property
    def shape(self):
        return (self.uid.shape[0], self.uid.shape[1], self.table.shape[1])

    def __getitem__(self, idx):
        if (isinstance(idx, tuple) and len(idx) == 2 and idx[0] == slice(None) and isinstance(idx[1], slice)
                and idx[1].start in (None, 0) and idx[1].step in (None, 1)):
            v = self.uid[:, idx[1]]
            dom = getattr(self.uid, "_kp_dom", None)
            if dom is not None:                      # (per-hop hint for the dictionary gradient: follows the hop prefix)
                v._kp_dom = dom[idx[1]]
            return DictPeripheral(self.table, v)
        return self.dense()[idx]

    def dense(self):
        """Materialised [N,K,D] tensor (differentiable): for callers outside the fused kernels."""
        return DictRows.apply(self.table, self.uid)


def aggregate_fwd_raw(csr, k_act, mode, x, table0, tablek, periph, eps, theta, xbias, want_pre, ptab=None, uid=None,
                      xs=None, alphas=None, bf16=False):
    """Launch kpgnn_aggregate_fwd.  x is [N,k,D], or None with xs = k per-hop [N,D] tensors (row stride shared).
    bf16: xs are bf16 rows and `pre` is returned as bf16 (KPGNN_STORE_BF16).
    Returns (out or hout, pre or None)."""
    lib = _lib.load()
    if x is not None:
        N, K, D = x.shape
    else:
        N, D = xs[0].shape
        K = len(xs)
    assert K == k_act and N == csr.N
    dev = (x if x is not None else xs[0]).device
    d = _lib.AggFwdDesc()
    d.N, d.K, d.D, d.K_csr, d.mode = N, K, D, csr.K, mode
    d.n_dyn = dyn_ptr(N)
    use_tables = table0 is not None
    d.use_tables = 1 if use_tables else 0
    d.n_code0 = table0.shape[0] if use_tables else 0
    d.n_codek = tablek.shape[0] if (use_tables and tablek is not None) else 0
    d.rowptr, d.col, d.code = csr.rowptr_dst.data_ptr(), csr.col_dst.data_ptr(), csr.code_dst.data_ptr()
    d.dis = csr.gcn_dis().data_ptr() if mode == MODE_GCN else None
    if x is not None:
        d.x, d.x_sn, d.x_sk = x.data_ptr(), x.stride(0), x.stride(1)
    else:
        d.x_sn = xs[0].stride(0)
        for k, t in enumerate(xs):
            assert t.shape == (N, D) and t.stride(1) == 1 and t.stride(0) == d.x_sn
            d.x_slot[k] = t.data_ptr()
    d.table0, d.tablek = _ptr(table0), _ptr(tablek)
    if periph is not None:
        d.periph, d.p_sn, d.p_sk = periph.data_ptr(), periph.stride(0), periph.stride(1)
    elif uid is not None:
        d.ptab, d.uid, d.uid_stride = ptab.data_ptr(), uid.data_ptr(), uid.stride(0)
        d.n_dict = ptab.shape[0]
    d.eps = _ptr(eps)
    d.xbias = _ptr(xbias)
    # dense K-hop neighbourhoods (>= 12 pairs per (node, hop) on average) of a batch whose graph boundaries are known: the
    # library may gather from an LDS-staged hop slab (mask-only aggregations; it checks the shape itself)
    gp = getattr(csr, "graph_ptr", None)
    if gp is not None and x is not None and not use_tables and csr.A >= 12 * max(csr.N * csr.K, 1) and d.n_dyn is None:
        d.graph_ptr, d.num_graphs, d.max_graph_nodes = gp.data_ptr(), gp.numel() - 1, csr.max_graph_nodes
    pre = torch.empty((N, K, D), dtype=torch.bfloat16 if bf16 else torch.float32, device=dev) if want_pre else None
    d.pre = _ptr(pre)
    d.storage = 1 if bf16 else 0
    if theta is not None:
        out = torch.empty((N, D), dtype=torch.float32, device=dev)
        d.theta, d.hout = theta.data_ptr(), out.data_ptr()
        d.alphas = _ptr(alphas)     # geometric combine: the launch fills `theta` itself
    else:
        out = torch.empty((N, K, D), dtype=torch.float32, device=dev)
        d.out, d.o_sn, d.o_sk = out.data_ptr(), out.stride(0), out.stride(1)
    with torch.cuda.device(dev):
        if _timer is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.kpgnn_aggregate_fwd(ctypes.byref(d), _stream(out)), "kpgnn_aggregate_fwd")
        if _timer is not None:
            e1.record()
            n_t = 1 + (periph is not None) + (pre is not None) + (theta is None)  # x, dense P, pre, out
            extra = 4 * N * K if uid is not None else 0                           # int32 uid per (node,hop)
            _timer.records.append(("agg_fwd", extra + algorithmic_bytes(csr, K, D, n_t, d.n_code0 + d.n_codek,
                                                                        extra_nd=1 if theta is not None else 0,
                                                                        s=2 if bf16 else 4), e0, e1))
    return out, pre


PULL_GATHER = True      # (tests switch it off to compare with the read-modify-write form of the backward gather)
_ones_cache = {}


def pull_applies(N, D, dtype=torch.float32):
    """Whether KHopAggregate.backward takes the pull form for a [N, k, D] gradient of this dtype (the KP-GIN+ history pattern
    is the caller's business): producers of other shares of a state's gradient use it to decide how to hand theirs over."""
    return PULL_GATHER and dtype == torch.float32 and N >= 4096 and D % 4 == 0


def khop_pull_gather(csr, slabs, hinit, hinit2=None):
    """A state's whole K-hop gradient in ONE launch (kpgnn_aggregate_fwd with the (source, hop)-keyed CSR, mode SUM, theta = 1
    and `hinit`): out[i] = hinit[i] + sum_k sum_{j in N_k(i)} slabs[k][j], where slabs[k] is hop k's [N,D] slab of dL/dS of the
    layer that read the state at hop slot k.  Replaces one read-modify-write of the state's gradient per reader (36 per step
    at K = L = 8: agg_bwd moved 273 MB per launch for 183 MB of gathered rows) by one write per state."""
    lib = _lib.load()
    K = len(slabs)
    N, D = slabs[0].shape
    dev = slabs[0].device
    key = (dev, D)
    ones = _ones_cache.get(key)
    if ones is None:
        ones = _ones_cache[key] = torch.ones((16, D), dtype=torch.float32, device=dev)
    out = hinit if hinit is not None else torch.empty((N, D), dtype=torch.float32, device=dev)
    d = _lib.AggFwdDesc()
    d.N, d.K, d.D, d.K_csr, d.mode = N, K, D, csr.K, MODE_SUM
    d.n_dyn = dyn_ptr(N)
    d.use_tables = 0
    d.rowptr, d.col, d.code = csr.rowptr_src.data_ptr(), csr.col_src.data_ptr(), csr.code_src.data_ptr()
    d.x_sn = D
    for k, t in enumerate(slabs):
        assert t.shape == (N, D) and t.is_contiguous() and t.dtype == torch.float32
        d.x_slot[k] = t.data_ptr()
    d.theta, d.hout, d.hinit, d.hinit2 = ones.data_ptr(), out.data_ptr(), _ptr(hinit), _ptr(hinit2)
    with torch.cuda.device(dev):
        if _timer is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.kpgnn_aggregate_fwd(ctypes.byref(d), _stream(out)), "kpgnn_aggregate_fwd (pull gather)")
        if _timer is not None:
            e1.record()
            _timer.records.append(("agg_bwd", algorithmic_bytes(csr, K, D, 1, 0, extra_nd=1 + (hinit is not None) + (hinit2 is not None)),
                                   e0, e1))
    return out


def aggregate_bwd_raw(csr, k_act, mode, g, eps, n_code0, n_codek, want_tables, slots=False, slot_bufs=None, gx_accum=None):
    """Launch kpgnn_aggregate_bwd on g = dL/dS.  Returns (gx, gtable0, gtablek); with slots=True gx is a list of
    k contiguous [N,D] tensors (one per hop slot) instead of one [N,k,D] tensor; slot_bufs[k] (a [N,D] tensor or None)
    makes the kernel ADD slot k's gradient into that buffer instead of writing a fresh one."""
    lib = _lib.load()
    N, K, D = g.shape
    dev = g.device
    bf16 = g.dtype == torch.bfloat16      # KPGNN_STORE_BF16: g rows are bf16, gx stays fp32
    d = _lib.AggBwdDesc()
    d.N, d.K, d.D, d.K_csr, d.mode = N, K, D, csr.K, mode
    d.n_dyn = dyn_ptr(N)
    d.storage = 1 if bf16 else 0
    d.use_tables = 1 if want_tables else 0
    d.n_code0, d.n_codek = n_code0, n_codek
    d.rowptr_src, d.col_src, d.code_src = csr.rowptr_src.data_ptr(), csr.col_src.data_ptr(), csr.code_src.data_ptr()
    d.dis = csr.gcn_dis().data_ptr() if mode == MODE_GCN else None
    d.g, d.g_sn, d.g_sk = g.data_ptr(), g.stride(0), g.stride(1)
    d.eps = _ptr(eps)
    late_adds = []
    if slots:
        # The kernel writes every hop slot with ONE row stride.  Parked gradient buffers (slot_bufs) may be strided
        # views (slices of the jumping-knowledge gradient, body.py): when all K slots are parked with the same stride
        # the kernel accumulates into them where they are; otherwise parked buffers whose stride differs from the fresh
        # contiguous outputs are added afterwards by the framework (rare path).
        bufs = list(slot_bufs) if slot_bufs is not None else [None] * K
        strides = {b.stride(0) for b in bufs if b is not None}
        sn = D
        if len(strides) == 1 and all(b is not None for b in bufs):
            sn = strides.pop()
        gx, mask = [], 0
        for k in range(K):
            b = bufs[k]
            if b is not None and b.stride(0) == sn and b.stride(1) == 1:
                gx.append(b)
                mask |= 1 << k
            else:
                t = torch.empty((N, D), dtype=torch.float32, device=dev)
                gx.append(t)
                if b is not None:
                    late_adds.append((k, b))
            d.gx_slot[k] = gx[k].data_ptr()
        d.gx_sn = sn
        d.accumulate_mask = mask
    elif gx_accum is not None:     # the state's gradient cell ([N, K*D] contiguous): the kernel adds into it
        gx = gx_accum.view(N, K, D)
        d.gx, d.gx_sn, d.gx_sk = gx.data_ptr(), gx.stride(0), gx.stride(1)
        d.accumulate_mask = (1 << K) - 1
    else:
        gx = torch.empty((N, K, D), dtype=torch.float32, device=dev)
        d.gx, d.gx_sn, d.gx_sk = gx.data_ptr(), gx.stride(0), gx.stride(1)
    gt0 = gtk = None
    if want_tables:
        gt0 = torch.zeros((n_code0, D), dtype=torch.float32, device=dev)
        d.gtable0 = gt0.data_ptr()
        if K > 1 and n_codek > 0:
            gtk = torch.zeros((n_codek, D), dtype=torch.float32, device=dev)
            d.gtablek = gtk.data_ptr()
    with torch.cuda.device(dev):
        if _timer is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.kpgnn_aggregate_bwd(ctypes.byref(d), _stream(g)), "kpgnn_aggregate_bwd")
        if _timer is not None:
            e1.record()
            # g read (2 or 4 bytes per element) + gx written (fp32)
            _timer.records.append(("agg_bwd", algorithmic_bytes(csr, K, D, 1.5 if bf16 else 2,
                                                                (n_code0 + n_codek) if want_tables else 0), e0, e1))
    for k, b in late_adds:
        gx[k] = b.add_(gx[k])
    return gx, gt0, gtk


def dict_tile_pack(csr, uid):
    """The (node, hop) dictionary entries of every tile of `csr`, sorted by dictionary row (kpgnn_dict_tile_pack): static
    per batch, built on first use and kept on the csr.  `uid` may be a hop-prefix view uid_full[:, :k]: the list is built
    over the full row (the kernel skips hops >= k).  Returns (pack, K_built) or (None, 0) when the shape has no pack."""
    kf = uid.stride(0) if uid.dim() == 2 and uid.shape[0] > 1 else uid.shape[1]
    if uid.stride(1) != 1 or kf > 8 or kf < uid.shape[1] or csr.nodes_per_tile * kf > 64:
        return None, 0
    cache = csr._dict_packs
    key = (uid.data_ptr(), kf, uid.shape[0])
    hit = cache.get(key)
    # (a StaticBatch refills `uid` in place: under dynamic_rows the list is rebuilt by every call - into the same buffer, so a
    #  captured graph refreshes it at every replay)
    if hit is None or dyn_ptr(uid.shape[0]) is not None:
        N = uid.shape[0]
        tiles = (N + csr.nodes_per_tile - 1) // csr.nodes_per_tile
        pack = hit[0] if hit is not None else torch.empty((tiles, 64), dtype=torch.int32, device=uid.device)
        with torch.cuda.device(uid.device):
            _lib.check(_lib.load().kpgnn_dict_tile_pack(uid.data_ptr(), kf, N, kf, csr.nodes_per_tile, pack.data_ptr(),
                                                        _stream(uid)), "kpgnn_dict_tile_pack")
        hit = cache[key] = (pack, kf, uid)     # (uid kept alive: the key is its address)
    return hit[0], hit[1]


def dict_grad_raw(uid, n_dict, theta, gh, defer=False):
    """Launch kpgnn_dict_grad: gdict[u] = sum_k theta[k] * sum_{i: uid[i,k]==u} gh[i].  uid [N,k] (may be a hop-prefix
    view), theta [k,D], gh [N,D].  Returns gdict [n_dict,D], or None when the shape does not fit the kernel.
    defer=True: returns (gdict, slab, nslab) with gdict still UNWRITTEN - the per-block partial sums wait in `slab` for a
    later finishing launch (table_grad_raw(extra=...) adds them up along with its own)."""
    lib = _lib.load()
    N, K = uid.shape
    D = gh.shape[1]
    ws_bytes = lib.kpgnn_dict_grad_workspace_bytes(N, K, D, n_dict)
    if ws_bytes == 0 or uid.stride(1) != 1:
        return None
    dev = gh.device
    dom = getattr(uid, "_kp_dom", None)          # [K] int32: the designated (most frequent) id per hop, if the caller has it
    has_dom = dom is not None and dom.numel() == K and dom.dtype == torch.int32 and dom.is_contiguous() and dom.device == dev
    if not has_dom and 4 * (n_dict * K * D + 2 * 64 * D) > 160 * 1024:
        return None        # (without the hint the kernel stages gh in LDS next to the accumulator rows: this dictionary does not fit)
    gh = gh.contiguous()
    theta = theta.contiguous()
    d = _lib.DictGradDesc()
    d.N, d.K, d.D, d.n_dict = N, K, D, n_dict
    d.n_dyn = dyn_ptr(N)
    d.uid, d.uid_stride, d.theta, d.gh = uid.data_ptr(), uid.stride(0), theta.data_ptr(), gh.data_ptr()
    if has_dom:
        d.dominant = dom.data_ptr()
    gd = torch.empty((n_dict, D), dtype=torch.float32, device=dev)
    ws = torch.empty(int(ws_bytes), dtype=torch.uint8, device=dev)
    d.gdict, d.workspace, d.workspace_bytes = gd.data_ptr(), ws.data_ptr(), int(ws_bytes)
    d.defer_reduce = 1 if defer else 0
    with torch.cuda.device(dev):
        if _timer is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.kpgnn_dict_grad(ctypes.byref(d), _stream(gh)), "kpgnn_dict_grad")
        if _timer is not None:
            e1.record()
            _timer.records.append(("dict_grad", 4 * N * D + 4 * N * K + 4 * D * (n_dict + K), e0, e1))
    if defer:
        return gd, ws, int(lib.kpgnn_dict_grad_slabs(N))
    return gd


DICT_MULTI = True      # one dictionary-gradient launch per backward pass instead of one per layer (where the layers qualify)


def dict_multi_ok(uid, n_dict, theta, gh):
    """Whether a layer's dictionary share can wait for the one launch of dict_grad_multi_raw."""
    if not DICT_MULTI or uid is None or theta is None:
        return False
    N, K = uid.shape
    D = gh.shape[1]
    dom = getattr(uid, "_kp_dom", None)
    return (dom is not None and dom.numel() == K and dom.dtype == torch.int32 and dom.is_contiguous() and dom.device == gh.device
            and uid.stride(1) == 1 and uid.dtype == torch.int32 and K <= 8 and D % 2 == 0 and D <= 128 and gh.dtype == torch.float32
            and theta.dtype == torch.float32 and tuple(theta.shape) == (K, D) and 4 * (n_dict * 8 + 9) * D <= 160 * 1024)


def dict_grad_multi_raw(items, n_dict):
    """kpgnn_dict_grad_multi over the parked (uid view, theta, gh) of the layers that read one dictionary:
    gdict[u] = sum_l sum_k theta_l[k] * sum_{i: uid[i,k]==u} gh_l[i].  The uid views are hop prefixes of ONE id matrix."""
    lib = _lib.load()
    uid = max((it[0] for it in items), key=lambda u: u.shape[1])
    N, K = uid.shape
    D = items[0][2].shape[1]
    dev = items[0][2].device
    base = uid.data_ptr()
    one = all(it[0].data_ptr() == base and it[0].stride(0) == uid.stride(0) and it[0].shape[0] == N for it in items)
    if not one or len(items) > 16 or 4 * (n_dict * K + 9 * len(items)) * D > 160 * 1024:
        # (different id matrices, more layers than the kernel takes, or a table + per-layer totals beyond the LDS: one launch per layer)
        total = None
        for u, th, gh in items:
            g = dict_grad_raw(u, n_dict, th, gh)
            if g is None:
                raise _lib.KpgnnError("kpgnn_dict_grad refused a share that was parked for kpgnn_dict_grad_multi")
            total = g if total is None else total + g
        return total
    keep = []
    d = _lib.DictGradMultiDesc()
    d.N, d.D, d.n_dict, d.L = N, D, n_dict, len(items)
    d.uid, d.uid_stride = base, uid.stride(0)
    for l, (u, th, gh) in enumerate(items):
        th, gh = th.contiguous(), gh.contiguous()
        keep += [th, gh]
        d.theta[l], d.gh[l], d.K[l] = th.data_ptr(), gh.data_ptr(), u.shape[1]
    d.dominant = uid._kp_dom.data_ptr()
    d.n_dyn = dyn_ptr(N)
    ws_bytes = int(lib.kpgnn_dict_grad_workspace_bytes(N, K, D, n_dict))
    gd = torch.empty((n_dict, D), dtype=torch.float32, device=dev)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    d.gdict, d.workspace, d.workspace_bytes = gd.data_ptr(), ws.data_ptr(), ws_bytes
    with torch.cuda.device(dev):
        if _timer is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.kpgnn_dict_grad_multi(ctypes.byref(d), _stream(gd)), "kpgnn_dict_grad_multi")
        if _timer is not None:
            e1.record()
            _timer.records.append(("dict_grad", len(items) * 4 * N * D + 4 * N * K + 4 * D * n_dict, e0, e1))
    return gd


def table_grad_raw(csr, g, n_code0, n_codek, edges=True, uid=None, n_dict=0, theta=None, gh=None, kernel=0, extra=None,
                   gdict_acc=None):
    """Launch kpgnn_table_grad on g = dL/dS [N,k,D]: edge-code table gradients (no per-edge atomics) and /
    or the peripheral-dictionary gradient (theta/gh given: sum theta[k]*gh[i]; else: sum of g rows).
    kernel: 0 automatic, 1 register walk, 2 count-matrix product (parity tests).
    Returns (gtable0, gtablek, gdict), or None when the shape fits neither kernel."""
    lib = _lib.load()
    g = g.contiguous()
    N, K, D = g.shape
    dev = g.device
    n0 = n_code0 if edges else 0
    nk = n_codek if (edges and K > 1) else 0
    ws_bytes = lib.kpgnn_table_grad_workspace_bytes(N, K, D, csr.nodes_per_tile, max(n0, 1) if edges else 0, nk, n_dict)
    if ws_bytes == 0:
        return None
    d = _lib.TableGradDesc()
    d.N, d.K, d.D, d.nodes_per_tile, d.n_code0, d.n_codek = N, K, D, csr.nodes_per_tile, n0, nk
    d.n_dyn = dyn_ptr(N)
    d.n_dict = n_dict
    d.dict_src = 0 if n_dict == 0 else (1 if theta is not None else 2)
    d.kernel = kernel
    d.storage = 1 if g.dtype == torch.bfloat16 else 0
    if edges:
        tptr, tpack = csr.tile_list(K)          # (hop-prefix copy of the entry list when this layer sees K < csr.K hops)
        d.tile_ptr, d.tile_pack = tptr.data_ptr(), tpack.data_ptr()
    d.g = g.data_ptr()
    d.g_sn, d.g_sk = K * D, D  # (contiguous; size-1 dims carry arbitrary strides)
    gt0 = gtk = gd = None
    if edges:
        gt0 = torch.empty((n0, D), dtype=torch.float32, device=dev)
        gtk = torch.empty((nk, D), dtype=torch.float32, device=dev) if nk > 0 else None
        d.gtable0, d.gtablek = gt0.data_ptr(), _ptr(gtk)
    if n_dict > 0:
        gd = torch.empty((n_dict, D), dtype=torch.float32, device=dev) if gdict_acc is None else gdict_acc
        d.uid, d.uid_stride, d.gdict = uid.data_ptr(), uid.stride(0), gd.data_ptr()
        if gdict_acc is not None:     # (a dictionary read by several layers: the gradient is added in place, see KHopAggregate)
            assert gdict_acc.is_contiguous() and gdict_acc.dtype == torch.float32 and tuple(gdict_acc.shape) == (n_dict, D)
            d.accumulate_dict = 1
        d.theta, d.gh = _ptr(theta), _ptr(gh)
        if K <= 8 and csr.nodes_per_tile * K <= 64:
            pack, kf = dict_tile_pack(csr, uid)
            d.dict_pack, d.dict_pack_K = _ptr(pack), kf
    ws = torch.empty(int(ws_bytes), dtype=torch.uint8, device=dev)
    d.workspace, d.workspace_bytes = ws.data_ptr(), int(ws_bytes)
    if extra is not None:       # (out, slab, nslab) of dict_grad_raw(defer=True): added up by this call's finishing launch
        d.extra_out, d.extra_slab, d.extra_nslab, d.extra_elems = extra[0].data_ptr(), extra[1].data_ptr(), extra[2], extra[0].numel()
    with torch.cuda.device(dev):
        if _timer is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.kpgnn_table_grad(ctypes.byref(d), _stream(g)), "kpgnn_table_grad")
        if _timer is not None:
            e1.record()
            _timer.records.append(("table_grad", g.element_size() * N * K * D + (4 * csr.active_pairs(K) if edges else 0)
                                   + 4 * D * (n0 + nk + n_dict) + (4 * N * K if n_dict else 0), e0, e1))
    return gt0, gtk, gd


def _combine_table_grad_ok(csr, pre, table_rows=0, dict_rows=0):
    """Shapes the fused combine + table-gradient kernel takes (kpgnn_table_grad with fuse_pre): fp32, K <= 8, even D <= 128,
    tiles of 8 nodes, and an LDS plan that fits - tile + (table rows + dictionary rows + 40 service rows) * D floats
    <= 160 KB (e.g. train_SR.py's max_pe_num = 1000 does not: those layers keep the separate kernels)."""
    N, K, D = pre.shape
    lds = 4 * (8 * K * D + (table_rows + 2 * dict_rows + 40) * D) + 64
    ok = (K <= 8 and D % 2 == 0 and D <= 128 and csr.nodes_per_tile == 8
          and getattr(csr, "tile_ptr", None) is not None and pre.is_contiguous() and N > 0 and lds <= 160 * 1024)
    if pre.dtype == torch.bfloat16:
        # bf16 S / dL/dS (KPGNN_STORE_BF16): the matrix-core variant only - no dictionary rows riding along (large batches, where
        # kpgnn_dict_grad takes them), <= 64 table rows, every entry multiplicity known to be below 64
        return ok and N >= 4096 and table_rows <= 64 and 1 <= csr.max_multiplicity() < 64
    return ok and pre.dtype == torch.float32


def combine_table_grad_raw(csr, pre, gh, theta, ptab, uid, n_code0, n_codek, want_gtheta, alphas=None, extra=None,
                           dict_rows=0, gdict_acc=None):
    """kpgnn_table_grad with the combine backward fused in (KP-GIN+ epilogue, fp32): computes g = theta[k]*gh[i]*gelu'(S[i,k]),
    the edge-code table gradients from it and (want_gtheta) the theta gradient / d/dalphas - `g` is written once and never read
    back for the tables.  Returns (g, gtheta or (gtheta, galphas) or None, gtable0, gtablek), or None when the shape has no
    fused kernel (the caller then runs combine_bwd_raw + table_grad_raw).
    dict_rows > 0: the dictionary gradient (theta[k]*gh[i] per (node, hop) id) rides along the walk and is returned as a
    fifth value (small batches, where a separate dict_grad launch costs more than it saves).
    gdict_acc: a [n_dict, D] buffer the dictionary gradient (in-walk or `extra`) is ADDED to instead of being written."""
    lib = _lib.load()
    N, K, D = pre.shape
    dev = pre.device
    nk = n_codek if K > 1 else 0
    if not _combine_table_grad_ok(csr, pre, n_code0 + nk, ptab.shape[0] if (ptab is not None and uid is not None) else 0):
        return None
    gd = None
    if dict_rows > 0 and pre.dtype == torch.bfloat16:
        return None                     # (bf16 storage: the matrix-core variant carries no dictionary rows)
    if dict_rows > 0:
        pack, kf = dict_tile_pack(csr, uid)
        if pack is None:
            return None
    ws_bytes = lib.kpgnn_table_grad_workspace_bytes(N, K, D, csr.nodes_per_tile, max(n_code0, 1), nk, dict_rows)
    if ws_bytes == 0:
        return None
    d = _lib.TableGradDesc()
    d.N, d.K, d.D, d.nodes_per_tile, d.n_code0, d.n_codek = N, K, D, csr.nodes_per_tile, n_code0, nk
    d.n_dyn = dyn_ptr(N)
    if dict_rows > 0:
        gd = torch.empty((dict_rows, D), dtype=torch.float32, device=dev) if gdict_acc is None else gdict_acc
        d.n_dict, d.dict_src, d.uid, d.uid_stride, d.gdict = dict_rows, 1, uid.data_ptr(), uid.stride(0), gd.data_ptr()
        d.dict_pack, d.dict_pack_K = pack.data_ptr(), kf
    if gdict_acc is not None:
        assert gdict_acc.is_contiguous() and gdict_acc.dtype == torch.float32 and (dict_rows > 0 or extra is not None)
        d.accumulate_dict = 1
    tptr, tpack = csr.tile_list(K)
    d.tile_ptr, d.tile_pack = tptr.data_ptr(), tpack.data_ptr()
    d.max_multiplicity = csr.max_multiplicity()      # (< 64: the matrix-core kernel may take the table gradients)
    gh = gh.contiguous()
    theta = theta.contiguous()
    d.theta, d.gh = theta.data_ptr(), gh.data_ptr()
    gt0 = torch.empty((n_code0, D), dtype=torch.float32, device=dev)
    gtk = torch.empty((nk, D), dtype=torch.float32, device=dev) if nk > 0 else None
    d.gtable0, d.gtablek = gt0.data_ptr(), _ptr(gtk)
    ws = torch.empty(int(ws_bytes), dtype=torch.uint8, device=dev)
    d.workspace, d.workspace_bytes = ws.data_ptr(), int(ws_bytes)
    # dL/dS leaves hop-major ([K][N][D]: the transposed gather of kpgnn_aggregate_bwd then reads one contiguous [N,D] slab
    # per hop, like the forward's hop slots, instead of rows K*D floats apart - 1.5x of its bytes reached HBM that way)
    g = torch.empty((K, N, D), dtype=pre.dtype, device=dev).permute(1, 0, 2)
    d.g_sn, d.g_sk = g.stride(0), g.stride(1)
    d.storage = 1 if pre.dtype == torch.bfloat16 else 0
    d.fuse_pre, d.fuse_g = pre.data_ptr(), g.data_ptr()
    if uid is not None and ptab is not None:
        d.fuse_ptab, d.fuse_uid, d.fuse_uid_stride, d.fuse_n_dict = ptab.data_ptr(), uid.data_ptr(), uid.stride(0), ptab.shape[0]
    gth = gal = fws = None
    if want_gtheta:
        gth = torch.empty((K, D), dtype=torch.float32, device=dev)
        nb = int(lib.kpgnn_table_grad_fuse_workspace_bytes(K, D))
        fws = torch.empty(nb, dtype=torch.uint8, device=dev)
        d.fuse_gtheta, d.fuse_workspace, d.fuse_workspace_bytes = gth.data_ptr(), fws.data_ptr(), nb
        if alphas is not None:
            gal = torch.empty_like(alphas)
            d.fuse_alphas, d.fuse_galphas = alphas.data_ptr(), gal.data_ptr()
    if extra is not None:
        eo = extra[0] if gdict_acc is None else gdict_acc
        assert eo.numel() == extra[0].numel()
        d.extra_out, d.extra_slab, d.extra_nslab, d.extra_elems = eo.data_ptr(), extra[1].data_ptr(), extra[2], extra[0].numel()
    waiting = take_reduce_job()        # (an earlier launch's deferred reduction rides along with the finishing launch)
    if waiting is not None:
        d.pending = ctypes.cast(ctypes.pointer(waiting[0]), ctypes.c_void_p)
    with torch.cuda.device(dev):
        if _timer is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.kpgnn_table_grad(ctypes.byref(d), _stream(pre)), "kpgnn_table_grad")
        if _timer is not None:
            e1.record()     # S read, g written, gh read, pair list, tables
            _timer.records.append(("combine_table_grad", 8 * N * K * D + 4 * N * D + 4 * csr.active_pairs(K)
                                   + 4 * D * (n_code0 + nk), e0, e1))
    if dict_rows > 0:
        return g, ((gth, gal) if gal is not None else gth), gt0, gtk, gd
    return g, ((gth, gal) if gal is not None else gth), gt0, gtk


def combine_bwd_raw(mode, pre, gout, theta, periph, ptab, uid, want_gtheta, want_gv, alphas=None):
    """Launch kpgnn_combine_bwd.  gout is gh [N,D] when theta is given, else dL/dout [N,K,D].
    Returns (g, gv or None, gtheta or None); with `alphas` (geometric combine) the third is (gtheta, galphas)."""
    lib = _lib.load()
    N, K, D = pre.shape
    dev = pre.device
    refuse_dynamic_rows("kpgnn_combine_bwd", N)
    d = _lib.CombineBwdDesc()
    d.N, d.K, d.D, d.mode = N, K, D, mode
    d.pre = pre.data_ptr()
    if theta is not None:
        d.gh, d.theta = gout.data_ptr(), theta.data_ptr()
    else:
        d.gout, d.go_sn, d.go_sk = gout.data_ptr(), gout.stride(0), gout.stride(1)
    if periph is not None:
        d.periph, d.p_sn, d.p_sk = periph.data_ptr(), periph.stride(0), periph.stride(1)
    elif uid is not None and ptab is not None:  # (P is only read for the theta gradient)
        d.ptab, d.uid, d.uid_stride = ptab.data_ptr(), uid.data_ptr(), uid.stride(0)
        d.n_dict = ptab.shape[0]
    bf16 = pre.dtype == torch.bfloat16    # KPGNN_STORE_BF16: S arrives and dL/dS leaves as bf16 rows
    d.storage = 1 if bf16 else 0
    g = torch.empty((N, K, D), dtype=pre.dtype, device=dev)
    gv = torch.empty((N, K, D), dtype=torch.float32, device=dev) if want_gv else None
    gth = torch.empty((K, D), dtype=torch.float32, device=dev) if want_gtheta else None
    d.g, d.gv, d.gtheta = g.data_ptr(), _ptr(gv), _ptr(gth)
    ws = gal = None
    if want_gtheta and alphas is not None:
        gal = torch.empty_like(alphas)
        d.alphas, d.galphas = alphas.data_ptr(), gal.data_ptr()
    if want_gtheta:
        nb = lib.kpgnn_combine_bwd_workspace_bytes(N, K, D)
        ws = torch.empty(int(nb), dtype=torch.uint8, device=dev)
        d.workspace, d.workspace_bytes = ws.data_ptr(), int(nb)
    with torch.cuda.device(dev):
        if _timer is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(lib.kpgnn_combine_bwd(ctypes.byref(d), _stream(pre)), "kpgnn_combine_bwd")
        if _timer is not None:
            e1.record()
            n_t = 2 + (gv is not None) + (theta is None) + (periph is not None and want_gtheta)
            _timer.records.append(("combine_bwd", (2 if bf16 else 4) * N * K * D * n_t + (4 * N * D if theta is not None else 0), e0, e1))
    if gal is not None:
        return g, gv, (gth, gal)
    return g, gv, gth


class KHopAggregate(torch.autograd.Function):
    """out[N,k,D] (or hout[N,D] with a fused geometric combine) = epilogue(K-hop segmented sum).

    Differentiable w.r.t. x, table0, tablek, periph (dense) or ptab (dictionary), eps, theta.  `xbias` (a
    detached constant row, see kpgnn.h) carries no gradient: it is the padding row of hopk_node_path_emb,
    whose grad the reference's nn.Embedding(padding_idx=0) discards as well.

    Backward = three HIP launches: kpgnn_combine_bwd (g = dL/dS, theta grad) when the epilogue has an
    activation or a fused combine, kpgnn_table_grad (edge-code table + dictionary grads, no atomics),
    kpgnn_aggregate_bwd (the transposed gather for dL/dx)."""

    @staticmethod
    def forward(ctx, x, table0, tablek, periph, eps, theta, xbias, ptab, csr, k_act, mode, uid, cells, *xs):
        # cells: per-hop inputs -> list of the slots' gradient cells; one [N,K,D] input -> the gradient cell of the STATE that x
        # is a view of (or None): its other readers' gradients are then collected in one buffer (see khop_aggregate)
        ctx.cells = cells if xs else None
        ctx.x_cell = cells if (not xs and cells is not None) else None
        _require_cuda(x, table0, tablek, periph, eps, theta, xbias, ptab, uid, *xs)
        ctx.n_slots = len(xs)
        bf16 = False
        if xs:   # per-hop inputs: k separate [N,D] states instead of one stacked [N,k,D] tensor
            assert x is None and len(xs) == k_act
            # bf16 storage (set_storage_dtype): only the configuration the bf16 kernels exist for - the fused KP-GIN+
            # epilogue with a dictionary P and code tables; everything else stays fp32
            bf16 = (_STORAGE is torch.bfloat16 and mode == MODE_GINPLUS and theta is not None and ptab is not None
                    and periph is None and table0 is not None and xs[0].shape[1] % 8 == 0 and eps is None and k_act > 1)
            # (k_act > 1: the single-hop first layer reads the raw input embedding once - nothing to save, and its rounding
            #  is what the ill-conditioned gradients of the input encoders' scalar gates feel first)
            xs = [bf16_shadow(t) if bf16 else t.float() for t in xs]
            if ctx.cells:   # will this reader's backward be the pull form?  (not with the unfused epilogue - the attention combine hands
                            #  back a node-major [N,k,D] gradient - and not with bf16 storage, whose dL/dS is bf16)
                ctx.cells[0].pull_reader = bool(PULL_GATHER and mode == MODE_GINPLUS and theta is not None and eps is None and not bf16)
            # the kernel reads every hop slot with ONE row stride (x_sn): row-strided slots (column slices of the bodies'
            # jumping-knowledge buffer) are read where they are; only mixed layouts are copied
            if any(t.stride(1) != 1 for t in xs) or len({t.stride(0) for t in xs}) != 1:
                xs = [t.contiguous() for t in xs]
        else:
            x = _last_contig(x.float())
        if periph is not None:
            periph = _last_contig(periph.float())
            ptab = uid = None
        # a dictionary every layer of a sequential stack reads (the bodies mark it, body._peripheral): its gradient is collected
        # in ONE buffer.  The first reader in forward order is the last to run backward: it hands autograd the total.
        ctx.dict_cell, ctx.dict_first = None, False
        if ptab is not None and getattr(ptab, "_kp_shared_grad", False):
            c = getattr(ptab, "_kp_grad_cell", None)
            ctx.dict_first = c is None
            if c is None:
                c = ptab._kp_grad_cell = _SlotGradCell()
            ctx.dict_cell = c
        if table0 is not None:
            table0 = table0.contiguous()
            tablek = tablek.contiguous() if tablek is not None else None
            _check_codes(csr, k_act, table0, tablek)
        alphas = None
        if theta is not None and theta.dim() == 1:
            # GeometricCombine.alphas [D]: theta = softmax_k(a (1-a)^k) is computed by the aggregation launch itself
            alphas = theta.contiguous()
            theta = torch.empty((k_act, alphas.numel()), dtype=torch.float32, device=alphas.device)
        elif theta is not None:
            theta = theta.contiguous()
        if ptab is not None:
            ptab = ptab.contiguous()
        need_pre = mode in (MODE_GINPLUS, MODE_GCN) or theta is not None
        out, pre = aggregate_fwd_raw(csr, k_act, mode, x, table0, tablek, periph, eps, theta, xbias, need_pre,
                                     ptab=ptab, uid=uid, xs=list(xs) if xs else None, alphas=alphas, bf16=bf16)
        ctx.alphas = alphas
        ctx.csr, ctx.k_act, ctx.mode, ctx.uid = csr, k_act, mode, uid
        ctx.has_tables = table0 is not None
        ctx.n_code0 = table0.shape[0] if table0 is not None else 0
        ctx.n_codek = tablek.shape[0] if tablek is not None else 0
        ctx.has_periph = periph is not None
        ctx.n_dict = ptab.shape[0] if ptab is not None else 0
        eps_needs = eps is not None and eps.requires_grad
        if eps_needs and xs:
            x = torch.stack(list(xs), dim=1)
        ctx.save_for_backward(pre, eps, theta, periph if theta is not None else None,
                              x if eps_needs else None, xbias if eps_needs else None,
                              ptab if theta is not None else None)
        return out

    @staticmethod
    def backward(ctx, gout):
        pre, eps, theta, periph, x_saved, xbias, ptab = ctx.saved_tensors
        mode, csr, k_act, uid = ctx.mode, ctx.csr, ctx.k_act, ctx.uid
        galphas_done = False
        fused = theta is not None
        need_act = mode in (MODE_GINPLUS, MODE_GCN)
        want_gperiph = ctx.has_periph and ctx.needs_input_grad[3]
        want_gdict = ctx.n_dict > 0 and ctx.needs_input_grad[7]
        gtheta = gperiph = gdict = acc = None
        acc_used = False
        dict_parked = False
        gout = gout.contiguous() if fused else _last_contig(gout)
        want_tables = ctx.has_tables and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        gt0 = gtk = None
        tables_in_gather = False
        done = False
        # --- KP-GIN+ with the fused geometric combine, dictionary P and edge-code tables: ONE kernel computes dL/dS from S and
        #     gh, writes it for the gather below and takes the table gradients from the LDS copy of each tile; the dictionary
        #     gradient comes from gh alone (dict_grad), its partial sums added up by the same finishing launch
        if (fused and mode == MODE_GINPLUS and want_tables and periph is None and not want_gperiph
                and _combine_table_grad_ok(csr, pre, ctx.n_code0 + (ctx.n_codek if k_act > 1 else 0), ctx.n_dict)):
            extra = dg = None
            if want_gdict and ctx.dict_cell is not None and ctx.dict_cell.buf is not None:
                acc = ctx.dict_cell.buf                       # later layers' share: the finishing launch adds to it
            if want_gdict and pre.shape[0] >= 4096:
                if ctx.dict_cell is not None and dict_multi_ok(uid, ctx.n_dict, theta, gout):
                    ctx.dict_cell.park_dict(uid, theta, gout)        # ONE launch for all the layers, run by the last of them
                    dict_parked = True
                    acc = None
                else:
                    dg = dict_grad_raw(uid, ctx.n_dict, theta, gout, defer=True)
                    extra = dg
            in_walk = ctx.n_dict if (want_gdict and dg is None and not dict_parked) else 0   # small batches: the dictionary rows ride along
            r = combine_table_grad_raw(csr, pre, gout, theta, ptab, uid, ctx.n_code0, ctx.n_codek,
                                       want_gtheta=ctx.needs_input_grad[5], alphas=ctx.alphas, extra=extra, dict_rows=in_walk,
                                       gdict_acc=acc)
            if r is None and in_walk:
                r = combine_table_grad_raw(csr, pre, gout, theta, ptab, uid, ctx.n_code0, ctx.n_codek,
                                           want_gtheta=ctx.needs_input_grad[5], alphas=ctx.alphas)
                in_walk = 0
            if r is None and dict_parked:            # (no fused kernel for this shape after all: the share goes the plain way below)
                ctx.dict_cell._dg.pop()
                dict_parked = False
            if r is not None:
                g, gtheta, gt0, gtk = r[:4]
                if isinstance(gtheta, tuple):
                    gtheta = gtheta[1]
                    galphas_done = True
                if dg is not None:
                    gdict = dg[0] if acc is None else acc
                    acc = None
                elif in_walk:
                    gdict = r[4]
                    acc = None
                elif want_gdict and not dict_parked:
                    gdict = table_grad_raw(csr, g, 0, 0, edges=False, uid=uid, n_dict=ctx.n_dict, theta=theta, gh=gout)[2]
                done = True
            elif dg is not None:
                raise _lib.KpgnnError("table_grad refused a shape after dict_grad deferred its reduction to it")
        if done:
            pass
        elif fused or need_act:
            g, gv, gtheta = combine_bwd_raw(mode, pre, gout, theta, periph, ptab, uid,
                                            want_gtheta=fused and ctx.needs_input_grad[5],
                                            want_gv=fused and want_gperiph, alphas=ctx.alphas if fused else None)
            if isinstance(gtheta, tuple):            # geometric combine: d/dalphas came with the same finishing launch
                gtheta = gtheta[1]
                galphas_done = True
            gperiph = gv if fused else (gout if want_gperiph else None)
        else:
            g = gout
            gperiph = gout if want_gperiph else None
        # --- table gradients (edge codes + peripheral dictionary), column-private kernel
        if not done and (want_tables or want_gdict):
            edges_here = want_tables and mode != MODE_GCN   # GCN weights its table grads per edge: fused atomics
            dict_here = want_gdict and (fused or not need_act)  # g == dL/dP only without an activation
            res = None
            extra = None
            # (below ~4K nodes the walk's own dictionary path is cheaper than a second kernel: 1.44 vs 1.52 ms per step at
            #  batch 64, where every launch is latency-bound)
            if dict_here and fused and (g.shape[0] >= 4096 or not edges_here):   # from gh alone (20 MB, not [N,K,D])
                # (its partial sums are added up by table_grad's finishing launch when one follows)
                dg = dict_grad_raw(uid, ctx.n_dict, theta, gout, defer=edges_here)
                if dg is not None:
                    dict_here = False
                    if edges_here:
                        gdict, extra = dg[0], dg
                    else:
                        gdict = dg
            if edges_here or dict_here:
                if dict_here and ctx.dict_cell is not None and ctx.dict_cell.buf is not None:
                    acc = ctx.dict_cell.buf          # later layers' share: this call's finishing launch adds to it
                res = table_grad_raw(csr, g, ctx.n_code0, ctx.n_codek, edges=edges_here,
                                     uid=uid if dict_here else None, n_dict=ctx.n_dict if dict_here else 0,
                                     theta=theta if fused else None, gh=gout if fused else None, extra=extra,
                                     gdict_acc=acc if dict_here else None)
                if res is None and extra is not None:
                    raise _lib.KpgnnError("table_grad refused a shape after dict_grad deferred its reduction to it")
                if res is not None and dict_here and acc is not None:
                    acc_used = True
            if res is not None:
                gt0, gtk = res[0], res[1]
                gdict = res[2] if gdict is None else gdict
            if want_tables and (res is None or not edges_here):
                tables_in_gather = True
            if want_gdict and gdict is None:  # activation without fused combine (or oversize tables): dL/dP = gout
                r2 = table_grad_raw(csr, gout if not fused else (gout.unsqueeze(1) * theta.unsqueeze(0)),
                                    0, 0, edges=False, uid=uid, n_dict=ctx.n_dict)
                if r2 is None:
                    raise _lib.KpgnnError("peripheral dictionary too large for the LDS table-gradient kernel; "
                                          "pass a dense peripheral_attr instead")
                gdict = r2[2]
        if ctx.dict_cell is not None and want_gdict and ctx.dict_first:
            items = ctx.dict_cell.take_dicts()       # the last reader in backward order: every parked layer's share in ONE launch
            if items:
                gm = dict_grad_multi_raw(items, ctx.n_dict)
                gdict = gm if gdict is None else gdict + gm
        if ctx.dict_cell is not None and want_gdict:
            if dict_parked and not ctx.dict_first:
                gdict = None                         # (this layer's share waits in the cell's list; a share in `buf` stays where it is)
            else:
                if acc_used:
                    acc = None                       # (already inside gdict)
                elif acc is None and (not done or dict_parked) and ctx.dict_cell.buf is not None:
                    acc = ctx.dict_cell.buf
                if acc is not None:                  # (a path without the accumulating launch: add the parked share the plain way)
                    gdict = gdict + acc
                if ctx.dict_first:
                    ctx.dict_cell.buf = None         # the total goes to autograd
                else:
                    ctx.dict_cell.buf, gdict = gdict, None
        xbuf = None
        if (ctx.x_cell is not None and ctx.x_cell.buf is not None and ctx.needs_input_grad[0] and k_act <= 32
                and mode != MODE_GCN and not tables_in_gather):
            b = ctx.x_cell.buf
            if b.is_contiguous() and b.numel() == g.numel() and b.dtype == torch.float32:
                xbuf = b
        # --- pull form (KP-GIN+ history pattern, large batches): this layer's hop slabs of g are parked with the states it read
        #     as slots >= 1; the state it read as slot 0 - whose last reader it is - gets its WHOLE gradient from one gather over
        #     the slabs later layers parked for it, this layer's hop-0 slab and whatever share the cell already holds
        pull = (PULL_GATHER and ctx.n_slots > 0 and ctx.cells is not None and mode == MODE_GINPLUS and not tables_in_gather and eps is None
                and g.dtype == torch.float32 and g.shape[0] >= 4096 and g.shape[2] % 4 == 0 and k_act <= 16
                and all(g[:, k].is_contiguous() for k in range(k_act)))
        if pull:
            for k in range(1, k_act):
                ctx.cells[k].park_slab(k, g[:, k])
            c0 = ctx.cells[0]
            pend = c0.take_slabs()
            top = max(pend) if pend else 0
            zeros = None
            slabs = [g[:, 0]]
            for h in range(1, top + 1):
                t = pend.get(h)
                if t is None:            # (a hop nobody parked - a pruned backward pass: gathered from zeros)
                    zeros = torch.zeros_like(slabs[0]) if zeros is None else zeros
                    t = zeros
                slabs.append(t)
            hb = c0.buf
            shape = tuple(slabs[0].shape)
            ok = lambda t: t is not None and t.is_contiguous() and t.dtype == torch.float32 and tuple(t.shape) == shape
            hinit = hb if ok(hb) else None
            ext = c0.take_addend()
            hinit2 = ext if ok(ext) else None
            total = khop_pull_gather(csr, slabs[:16], hinit, hinit2)
            if hb is not None and hinit is None:
                total = total + hb.view_as(total)          # (odd layout: add the parked share the plain way)
            if ext is not None and hinit2 is None:
                total = total + ext.view_as(total)
            c0.buf = None
            return (None, gt0, gtk, gperiph, None, _finish_gtheta(ctx, gtheta, theta, galphas_done, k_act), None, gdict, None, None, None,
                    None, None, total, *([None] * (k_act - 1)))
        gx, a0, ak = aggregate_bwd_raw(csr, k_act, mode, g, eps, ctx.n_code0, ctx.n_codek, tables_in_gather,
                                       slots=ctx.n_slots > 0, slot_bufs=_slot_bufs(ctx), gx_accum=xbuf)
        if ctx.n_slots > 0 and ctx.cells is not None:
            ext = ctx.cells[0].take_addend()      # (a share handed over for the pull form that this layer could not run)
            if ext is not None:
                gx = list(gx)
                gx[0] = gx[0] + ext.view_as(gx[0])
            pend = ctx.cells[0].take_slabs()      # (later layers ran the pull form, this one could not: their slabs for ITS slot-0 state)
            if pend:
                zeros = torch.zeros_like(next(iter(pend.values())))
                extra = khop_pull_gather(csr, [pend.get(h, zeros) for h in range(0, max(pend) + 1)], None)
                gx = list(gx)
                gx[0] = gx[0] + extra
        if ctx.x_cell is not None:
            if xbuf is None and ctx.x_cell.buf is not None and ctx.needs_input_grad[0]:
                gx = gx + ctx.x_cell.buf.view_as(gx)     # (odd layout: add the parked share the plain way)
            ctx.x_cell.buf = None
        if tables_in_gather:
            gt0, gtk = a0, ak
        gtheta = _finish_gtheta(ctx, gtheta, theta, galphas_done, k_act)
        geps = None
        if eps is not None and ctx.needs_input_grad[4] and mode == MODE_GIN:
            refuse_dynamic_rows("the eps gradient (framework sum over rows)", g.shape[0])
            xe = x_saved
            if xbias is not None and xe.shape[1] > 1:
                xe = torch.cat([xe[:, :1], xe[:, 1:] + xbias], dim=1)
            geps = (g * xe).sum().reshape(eps.shape)
        # (row 0 of both table grads stays exactly zero: code 0 == "inactive" never enters the CSR, which
        #  is also what nn.Embedding(padding_idx=0) would do)
        if ctx.n_slots:
            return (None, gt0, gtk, gperiph, geps, gtheta, None, gdict, None, None, None, None, None,
                    *_slot_grads(ctx, gx))
        return (gx if ctx.needs_input_grad[0] else None, gt0, gtk, gperiph, geps, gtheta, None, gdict,
                None, None, None, None, None)


def _finish_gtheta(ctx, gtheta, theta, galphas_done, k_act):
    """d/dalphas through theta (geo_theta.hip) when the finishing launch has not already produced it."""
    if gtheta is not None and ctx.alphas is not None and not galphas_done:
        galpha = torch.empty_like(ctx.alphas)
        lib = _lib.load()
        with torch.cuda.device(galpha.device):
            _lib.check(lib.kpgnn_geo_theta_bwd(ctx.alphas.data_ptr(), theta.data_ptr(), gtheta.data_ptr(), k_act,
                                               ctx.alphas.numel(), galpha.data_ptr(), _stream(galpha)), "kpgnn_geo_theta_bwd")
        return galpha
    return gtheta


def _backward_pass_id():
    """Identity of the autograd backward pass that is executing (-1 outside one)."""
    try:
        return torch._C._current_graph_task_id()
    except AttributeError:  # pragma: no cover - older torch
        return -1


class _SlotGradCell:
    """Gradient buffer of one state tensor that several layers read as a hop slot (see khop_aggregate).

    Cells carry part of d/dstate OUTSIDE autograd: a reader parks its share in `buf`, later readers (in backward order) add
    to it in place and the state's last reader hands autograd the total and clears the cell.  That is only sound within ONE
    backward pass.  A pruned or partial pass (`autograd.grad(score, some_params, retain_graph=True)`) may park a share that
    no reader of ITS pass ever collects; the parked buffer is therefore tagged with the pass that wrote it
    (torch._C._current_graph_task_id()) and reads from any other pass see an empty cell - a stale share is never added to
    a later pass's gradient (tests/test_gpu_parity.py::test_gradient_cells_survive_a_partial_backward)."""
    __slots__ = ("_buf", "_task", "_pend", "_ptask", "_add", "_atask", "pull_reader", "_dg", "_dgtask")

    def __init__(self):
        self._buf, self._task = None, -1
        self._pend, self._ptask = None, -1
        self._add, self._atask = None, -1       # one more [N,D] addend of the pull gather (the residual branch's share)
        self.pull_reader = False                # set in forward by the state's slot-0 reader when its backward will be the pull
                                                # form (fused geometric combine): only then is an addend worth parking
        self._dg, self._dgtask = None, -1       # dictionary cell: the (uid, theta, gh) of the layers whose share waits for ONE launch

    # One dictionary-gradient launch for all the layers that read the dictionary (dict_grad_multi_raw): every reader parks
    # (uid view, theta, gh); the first reader in forward order - the last to run backward - runs the launch.  Pass-tagged like `buf`.
    def park_dict(self, uid, theta, gh):
        if self._dg is None or self._dgtask != _backward_pass_id():
            self._dg, self._dgtask = [], _backward_pass_id()
        self._dg.append((uid, theta, gh))

    def take_dicts(self):
        t = self._dg if self._dgtask == _backward_pass_id() else None
        self._dg, self._dgtask = None, -1
        return t or []

    def park_addend(self, t):
        self._add, self._atask = t, _backward_pass_id()

    def take_addend(self):
        t = self._add if self._atask == _backward_pass_id() else None
        self._add, self._atask = None, -1
        return t


    # Pull form of the backward gather (khop_pull_gather): a later reader of the state does not add its share into `buf` -
    # it leaves the hop slab of ITS dL/dS here, keyed by the hop slot it read the state at, and the state's last reader
    # gathers from all of them in one launch.  Same pass tagging as `buf`.
    def park_slab(self, hop, slab):
        if self._pend is None or self._ptask != _backward_pass_id():
            self._pend, self._ptask = {}, _backward_pass_id()
        self._pend[hop] = slab

    def take_slabs(self):
        p = self._pend if (self._pend is not None and self._ptask == _backward_pass_id()) else {}
        self._pend, self._ptask = None, -1
        return p

    @property
    def buf(self):
        if self._buf is not None and self._task != _backward_pass_id():
            self._buf = None                # parked by another backward pass: stale
        return self._buf

    @buf.setter
    def buf(self, v):
        self._buf = v
        self._task = _backward_pass_id() if v is not None else -1


def state_cell(t):
    """The gradient cell of a state tensor (created on first use): whoever computes part of d/dt BEFORE the state's last
    reader runs its backward may add it to `cell.buf` instead of handing autograd a tensor to sum."""
    c = getattr(t, "_kp_slot_cell", None)
    if c is None:
        c = t._kp_slot_cell = _SlotGradCell()
    return c


def _slot_bufs(ctx):
    return [c.buf for c in ctx.cells] if ctx.cells else None


def _slot_grads(ctx, gx):
    """With shared cells: slots >= 1 park their gradient in the state's cell (the kernel has added to it) and hand
    autograd nothing; slot 0 - by construction the LAST of a state's readers to run backward - returns the total."""
    if not ctx.cells:
        return gx
    out = []
    for k, c in enumerate(ctx.cells):
        if k == 0:
            out.append(gx[0])
            c.buf = None
        else:
            c.buf = gx[k]
            out.append(None)
    return out


def khop_aggregate(x, csr, k_act, mode, table0=None, tablek=None, periph=None, eps=None, theta=None, xbias=None,
                   share_slot_grads=False, x_state=None):
    """x: [N,k,D] tensor, or a list/tuple of k per-hop [N,D] tensors (no stacking copy).
    periph: dense [N,k,D] tensor, a DictPeripheral, or None.
    theta: [k,D] hop weights of a fused combine, or the [D] `alphas` of a GeometricCombine (theta is then computed by the
    aggregation launch itself and differentiated back to alphas).
    share_slot_grads (per-hop list only): the caller guarantees the GNNPlus history pattern - slot k of layer m is the
    output of layer m-1-k, so every state is read as slot 0 by the layer right after it and as slots >= 1 only by LATER
    layers, whose backward autograd runs first.  The backward then accumulates a state's slot gradients in ONE buffer
    inside the gather kernel (kpgnn_agg_bwd_desc.accumulate_mask) instead of emitting one [N,D] tensor per reader
    for autograd to add up (36 adds of 19.7 MB per step at K = L = 8)."""
    ptab = uid = None
    if isinstance(periph, DictPeripheral):
        ptab, uid, periph = periph.table, periph.uid, None
    if isinstance(x, (list, tuple)):
        cells = None
        if share_slot_grads and torch.is_grad_enabled():
            cells = []
            for t in x:
                c = getattr(t, "_kp_slot_cell", None)
                if c is None:
                    c = t._kp_slot_cell = _SlotGradCell()
                cells.append(c)
        return KHopAggregate.apply(None, table0, tablek, periph, eps, theta, xbias, ptab, csr, k_act, mode, uid, cells, *x)
    # x_state: the [N, K*D] state tensor that x is a view of, when the caller guarantees that this call is the LAST of the
    # state's readers to run backward (the bodies: the jumping-knowledge projection and the next norm's residual branch read
    # it too, both later in the forward).  Those readers park their share of d/dstate in the state's cell and the gather
    # kernel adds its own into the same buffer - no [N,H] tensors for autograd to sum.
    cell = None
    if x_state is not None and torch.is_grad_enabled() and x_state.requires_grad and x_state.is_cuda:
        cell = state_cell(x_state)
    return KHopAggregate.apply(x, table0, tablek, periph, eps, theta, xbias, ptab, csr, k_act, mode, uid, cell)


class DictRows(torch.autograd.Function):
    """out[i,k,:] = table[uid[i,k],:]; backward through kpgnn_table_grad (dictionary part, no atomics)."""

    @staticmethod
    def forward(ctx, table, uid):
        ctx.save_for_backward(uid)
        ctx.n = table.shape[0]
        return table.index_select(0, uid.reshape(-1).long()).view(uid.shape[0], uid.shape[1], table.shape[1])

    @staticmethod
    def backward(ctx, gout):
        (uid,) = ctx.saved_tensors
        gout = _last_contig(gout)

        class _C:  # table_grad_raw only needs the tile size from the CSR when there is no pair list
            nodes_per_tile = 8 if gout.shape[1] <= 8 else max(1, 64 // gout.shape[1])
        res = table_grad_raw(_C, gout, 0, 0, edges=False, uid=uid, n_dict=ctx.n)
        if res is None:
            g = torch.zeros((ctx.n, gout.shape[2]), dtype=gout.dtype, device=gout.device)
            return g.index_add_(0, uid.reshape(-1).long(), gout.reshape(-1, gout.shape[2])), None
        return res[2], None


# ------------------------------------------------------------------------------------------------ table gather-sum
class TableGatherSum(torch.autograd.Function):
    """out[m,:] = bias + sum_c table[col_offset[c] + idx[m,c], :]  (kpgnn_table_gather_sum_fwd/bwd)."""

    @staticmethod
    def forward(ctx, table, bias, idx, col_offset):
        _require_cuda(table, bias, idx, col_offset)
        lib = _lib.load()
        table = table.contiguous()
        bias = bias.contiguous() if bias is not None else None
        M, C = idx.shape
        R, D = table.shape
        out = torch.empty((M, D), dtype=torch.float32, device=table.device)
        d = _lib.TgsDesc()
        d.M, d.C, d.D, d.R = M, C, D, R
        d.n_dyn = dyn_ptr(M)
        d.idx, d.col_offset, d.table, d.bias = idx.data_ptr(), col_offset.data_ptr(), table.data_ptr(), _ptr(bias)
        d.out, d.out_stride = out.data_ptr(), out.stride(0)
        with torch.cuda.device(table.device):
            _lib.check(lib.kpgnn_table_gather_sum_fwd(ctypes.byref(d), _stream(table)), "kpgnn_table_gather_sum_fwd")
        ctx.save_for_backward(idx, col_offset)
        ctx.shape = (R, D)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, gout):
        idx, col_offset = ctx.saved_tensors
        lib = _lib.load()
        gout = _last_contig(gout)
        R, D = ctx.shape
        M, C = idx.shape
        gtable = torch.empty((R, D), dtype=torch.float32, device=gout.device)
        nb = int(lib.kpgnn_table_gather_sum_bwd_workspace_bytes(M, D, R))
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=gout.device)
        d = _lib.TgsDesc()
        d.M, d.C, d.D, d.R = M, C, D, R
        d.n_dyn = dyn_ptr(M)
        d.idx, d.col_offset = idx.data_ptr(), col_offset.data_ptr()
        d.gout, d.gout_stride, d.gtable = gout.data_ptr(), gout.stride(0), gtable.data_ptr()
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        with torch.cuda.device(gout.device):
            _lib.check(lib.kpgnn_table_gather_sum_bwd(ctypes.byref(d), _stream(gout)), "kpgnn_table_gather_sum_bwd")
        if ctx.has_bias:
            refuse_dynamic_rows("the bias gradient of a table gather-sum (framework row sum)", M)
        gbias = gout.sum(0) if ctx.has_bias else None
        return gtable, gbias, None, None


def table_gather_sum(table, bias, idx, col_offset):
    return TableGatherSum.apply(table, bias, idx, col_offset)


_I16 = "_kpgnn_idx16"
_ZERO_OFF = {}


def _zero_offset(dev):
    """One int32 zero per device: the column offset of a single-table gather-sum."""
    t = _ZERO_OFF.get(dev)
    if t is None:
        t = _ZERO_OFF[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    return t


class _ZeroRowGrad(torch.autograd.Function):
    """Identity whose backward zeroes one row of the gradient: nn.Embedding(padding_idx=r) semantics for a table that
    is read through the gather-sum kernels."""

    @staticmethod
    def forward(ctx, weight, row):
        ctx.row = row
        return weight.view_as(weight)

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        g[ctx.row].zero_()
        return g, None


def embedding_rows(weight, idx, padding_idx=None):
    """weight[idx] for an integer index tensor of any shape (the bodies' input embedding, input_encoder.py:21-22; the
    layers' path encoding, KPGIN.py:92-93) through the gather-sum kernels: unlike the framework's embedding backward
    (sort + unique_by_key with a host read-back, which FAULTS when a captured hipGraph replays it) this is free of
    host synchronisation.  The 16-bit copy of the index is cached on the index tensor object; building it validates
    the range once (one sync per index tensor, outside any capture), as nn.Embedding's device assert would.
    padding_idx: that row receives no gradient (its forward value is whatever the table holds, zeros for nn.Embedding)."""
    if idx.dtype not in (torch.int64, torch.int32, torch.int16, torch.uint8):
        raise TypeError(f"embedding_rows: integer index expected, got {idx.dtype}")
    R = weight.shape[0]
    if R > 65536:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.KpgnnError(f"embedding table with {R} rows > 65536 falls back to the framework's embedding, whose "
                                  "backward cannot be captured in a hipGraph (host read-back); run eagerly")
        return torch.nn.functional.embedding(idx.long(), weight, padding_idx=padding_idx)
    rec = getattr(idx, _I16, None)
    if rec is None or rec[0] != (idx._version, R):
        # a batch collated from a resident dataset (dataset.KHopDataset.collate) carries int16 indices whose dataset-wide
        # maximum is known on the host: no device round trip for the range check
        pre = getattr(idx, "_kp_index_bound", None)
        if pre is not None and pre[0] == idx._version and idx.dtype == torch.int16 and idx.is_contiguous():
            if pre[1] >= R:
                raise IndexError(f"embedding index {pre[1]} out of range for a table with {R} rows")
            rec = ((idx._version, R), idx.reshape(-1, 1), _zero_offset(idx.device))
            try:
                setattr(idx, _I16, rec)
            except Exception:  # pragma: no cover
                pass
    if rec is None or rec[0] != (idx._version, R):
        if torch.cuda.is_current_stream_capturing():
            raise _lib.KpgnnError("embedding_rows: first use of an index tensor inside a hipGraph capture (its range "
                                  "check needs a host sync); run one eager step on the batch before capturing")
        flat = idx.reshape(-1, 1)
        if flat.numel() and bool(((flat < 0) | (flat >= R)).any().item()):
            raise IndexError(f"embedding index out of range for a table with {R} rows")
        i64 = flat.long()
        i16 = (i64 if R <= 32768 else i64 - 65536 * (i64 >= 32768)).to(torch.int16).contiguous()
        off = torch.zeros(1, dtype=torch.int32, device=idx.device)
        rec = ((idx._version, R), i16, off)
        try:
            setattr(idx, _I16, rec)
        except Exception:  # pragma: no cover
            pass
    if padding_idx is not None and weight.requires_grad:
        weight = _ZeroRowGrad.apply(weight, int(padding_idx))
    res = TableGatherSum.apply(weight, None, rec[1], rec[2])
    return res.view(*idx.shape, weight.shape[1])


# ------------------------------------------------------------------------------------------------ graph readout
_GPTR = "_kpgnn_graph_ptr"


def graph_ptr_of(batch, num_graphs):
    """int32 [G+1] node offsets of the graphs of a collated batch (nodes of a graph are contiguous, as PyG's collate
    lays them out), cached on the `batch` tensor; checks once that `batch` is sorted (one sync per batch object)."""
    rec = getattr(batch, _GPTR, None)
    if rec is not None and rec[0] == (batch._version, num_graphs):
        return rec[1]
    if torch.cuda.is_current_stream_capturing():
        raise _lib.KpgnnError("graph readout: first use of a batch vector inside a hipGraph capture (its order check needs a "
                              "host sync); run one eager step on the batch before capturing")
    if batch.numel() > 1 and bool((batch[1:] < batch[:-1]).any().item()):
        raise ValueError("graph readout needs the nodes of every graph to be contiguous (sorted `batch`)")
    if batch.numel() and (int(batch[-1].item()) >= num_graphs or int(batch[0].item()) < 0):
        raise IndexError(f"graph readout: batch holds graph ids outside [0, {num_graphs})")
    ptr = torch.searchsorted(batch, torch.arange(num_graphs + 1, device=batch.device, dtype=batch.dtype)).to(torch.int32)
    try:
        setattr(batch, _GPTR, ((batch._version, num_graphs), ptr))
    except Exception:  # pragma: no cover
        pass
    return ptr


class SegmentPool(torch.autograd.Function):
    """out[g] = sum / mean of the rows of graph g (kpgnn_segment_pool_*): one launch per direction, no atomics."""

    @staticmethod
    def forward(ctx, x, batch, ptr, num_graphs, mean):
        lib = _lib.load()
        x = _last_contig(x)
        N, D = x.shape
        out = torch.empty((num_graphs, D), dtype=torch.float32, device=x.device)
        d = _lib.PoolDesc()
        d.N, d.G, d.D, d.mode = N, num_graphs, D, 1 if mean else 0
        d.n_dyn = dyn_ptr(N)
        d.graph_ptr, d.x, d.x_stride, d.out = ptr.data_ptr(), x.data_ptr(), x.stride(0), out.data_ptr()
        with torch.cuda.device(x.device):
            _lib.check(lib.kpgnn_segment_pool_fwd(ctypes.byref(d), _stream(x)), "kpgnn_segment_pool_fwd")
        ctx.save_for_backward(batch, ptr)
        ctx.dims = (N, D, num_graphs, mean)
        return out

    @staticmethod
    def backward(ctx, gout):
        batch, ptr = ctx.saved_tensors
        N, D, G, mean = ctx.dims
        lib = _lib.load()
        gout = gout.contiguous()
        gx = torch.empty((N, D), dtype=torch.float32, device=gout.device)
        d = _lib.PoolDesc()
        d.N, d.G, d.D, d.mode = N, G, D, 1 if mean else 0
        d.n_dyn = dyn_ptr(N)
        d.graph_ptr, d.batch, d.gout, d.gx, d.gx_stride = ptr.data_ptr(), batch.data_ptr(), gout.data_ptr(), gx.data_ptr(), D
        with torch.cuda.device(gout.device):
            _lib.check(lib.kpgnn_segment_pool_bwd(ctypes.byref(d), _stream(gout)), "kpgnn_segment_pool_bwd")
        return gx, None, None, None, None


def segment_pool(x, batch, num_graphs, mean=False):
    """Sum / mean readout of [N,D] node rows per graph on the HIP kernels (fp32 device tensors, int64 sorted batch)."""
    _require_cuda(x, batch)
    if x.shape[1] > 256:       # (kpgnn_segment_pool_* instantiates row widths up to 256 floats: wider rows take the framework's scatter)
        refuse_dynamic_rows("graph readout of rows wider than 256", x.shape[0])
        out = x.new_zeros((num_graphs, x.shape[1])).index_add_(0, batch.long(), x)
        if mean:
            cnt = x.new_zeros(num_graphs).index_add_(0, batch.long(), x.new_ones(batch.numel()))
            out = out / cnt.clamp(min=1).unsqueeze(-1)
        return out
    return SegmentPool.apply(x, batch.long() if batch.dtype != torch.int64 else batch, graph_ptr_of(batch, num_graphs), num_graphs, mean)


# ------------------------------------------------------------------------------------------------ projected tables
class EncTables(torch.autograd.Function):
    """(table [R,H], bias [H]) of the projected peripheral-feature tables (kpgnn_enc_tables_*): for encoder e with
    components c:  table[rows of c] = squash(gate_e) * Emb_c.weight @ W_e[:, cH:(c+1)H]^T,  bias = sum_e squash(gate_e) *
    mult_e * b_e.  One launch per direction.  Arguments: squash kind (0 sigmoid / 1 tanh), the encoders' multiplicities and
    component counts, then per encoder (proj.weight, proj.bias, gate) and the embedding weights encoder by encoder."""

    @staticmethod
    def forward(ctx, squash, mults, ncomps, *tensors):
        lib = _lib.load()
        ne = len(ncomps)
        enc_t, embs = tensors[:3 * ne], tensors[3 * ne:]
        _require_cuda(*tensors)
        H = embs[0].shape[1]
        dev = embs[0].device
        embs = [e.contiguous() for e in embs]
        enc_t = [t.contiguous() for t in enc_t]
        R = sum(e.shape[0] for e in embs)
        out = torch.empty((2, R, H), dtype=torch.float32, device=dev)       # table, pre
        bias = torch.empty((H,), dtype=torch.float32, device=dev)
        d = EncTables._desc(squash, mults, ncomps, enc_t, embs, H)
        d.table, d.pre, d.bias = out[0].data_ptr(), out[1].data_ptr(), bias.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_enc_tables_fwd(ctypes.byref(d), _stream(bias)), "kpgnn_enc_tables_fwd")
        ctx.save_for_backward(out, *enc_t, *embs)
        ctx.meta = (squash, tuple(mults), tuple(ncomps), H)
        return out[0], bias

    @staticmethod
    def _desc(squash, mults, ncomps, enc_t, embs, H):
        d = _lib.EncTablesDesc()
        d.H, d.num_components, d.num_encoders = H, len(embs), len(ncomps)
        c = 0
        for e, n in enumerate(ncomps):
            w, b, gate = enc_t[3 * e:3 * e + 3]
            assert tuple(w.shape) == (H, n * H) and b.numel() == H and gate.numel() == 1
            d.enc_w[e], d.enc_b[e], d.enc_gate[e] = w.data_ptr(), b.data_ptr(), gate.data_ptr()
            d.enc_mult[e], d.enc_squash[e] = float(mults[e]), int(squash)
            for _ in range(n):
                assert embs[c].shape[1] == H
                d.comp_emb[c], d.comp_rows[c], d.comp_encoder[c] = embs[c].data_ptr(), embs[c].shape[0], e
                c += 1
        assert c == len(embs)
        return d

    @staticmethod
    def backward(ctx, gtable, gbias):
        out, *rest = ctx.saved_tensors
        squash, mults, ncomps, H = ctx.meta
        ne = len(ncomps)
        enc_t, embs = rest[:3 * ne], rest[3 * ne:]
        lib = _lib.load()
        dev = out.device
        gtable = gtable.contiguous() if gtable is not None else torch.zeros_like(out[0])
        gbias = gbias.contiguous() if gbias is not None else torch.zeros(H, dtype=torch.float32, device=dev)
        d = EncTables._desc(squash, mults, ncomps, enc_t, embs, H)
        d.pre, d.gtable, d.gbias = out[1].data_ptr(), gtable.data_ptr(), gbias.data_ptr()
        g_enc = []
        for e in range(ne):
            gw, gb, gg = torch.empty_like(enc_t[3 * e]), torch.empty_like(enc_t[3 * e + 1]), torch.empty_like(enc_t[3 * e + 2])
            d.enc_gw[e], d.enc_gb[e], d.enc_ggate[e] = gw.data_ptr(), gb.data_ptr(), gg.data_ptr()
            g_enc += [gw, gb, gg]
        g_emb = [torch.empty_like(e) for e in embs]
        for c, g in enumerate(g_emb):
            d.comp_gemb[c] = g.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_enc_tables_bwd(ctypes.byref(d), _stream(gtable)), "kpgnn_enc_tables_bwd")
        return (None, None, None, *g_enc, *g_emb)


def enc_tables(squash, encoders):
    """encoders: list of (proj_weight [H, C*H], proj_bias [H], gate_raw [1], multiplicity, [Emb_c.weight ...]).
    Returns (table [R,H], bias [H])."""
    mults = [m for (_, _, _, m, _) in encoders]
    ncomps = [len(embs) for (_, _, _, _, embs) in encoders]
    enc_t = [t for (w, b, g, _, _) in encoders for t in (w, b, g)]
    embs = [e for (_, _, _, _, es) in encoders for e in es]
    return EncTables.apply(squash, mults, ncomps, *enc_t, *embs)
