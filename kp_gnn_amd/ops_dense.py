"""Dense tail of the layers: training-mode BatchNorm1d (+ReLU, +residual) on the HIP kernels of bn.hip.

The nn.BatchNorm1d modules stay where the reference has them (state_dict keys `mlp.1.*`, `norms.l.module.*`);
only their training-mode arithmetic is routed here.  Eval mode (running statistics) uses torch's own GPU op."""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib
from .ops import _ptr, _stream, dyn_ptr, refuse_dynamic_rows


# ------------------------------------------------------------------------------------------- column-statistics slots
STAT_REPLICAS = 8          # KPGNN_STAT_REPLICAS (include/kpgnn.h)
_EAGER_DOUBLES = 1 << 20   # 8 MB of slots between two zero fills when launching eagerly
_GRAPH_DOUBLES = 1 << 18   # 2 MB per captured graph (zero-filled by a node of the graph itself at every replay)


class _Arena:
    __slots__ = ("buf", "off")

    def __init__(self, n, dev):
        self.buf = torch.zeros(n, dtype=torch.float64, device=dev)
        self.off = 0


_arenas = {}   # device index -> {"eager": _Arena, "cap_id": int, "cap": _Arena}


def take_stat_slot(C, dev):
    """A zeroed column-statistics slot (double[STAT_REPLICAS][2][C], include/kpgnn.h) for ONE producer launch + ONE
    consumer launch.  Slots come from a per-device arena and are never handed out twice between two zero fills of the
    arena, so no per-use memset exists: eagerly the arena is re-zeroed every ~600 slots; a hipGraph capture gets an
    arena of its own, allocated and zero-filled INSIDE the capture (the fill is a node of the graph: every replay
    starts from zeros, and the memory lives in the graph's pool).
    Callers take every slot of an operator before they launch its first kernel (a wrap-around fill must not land
    between a slot's producer and its consumer)."""
    lib = _lib.load()
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _arenas.get(idx)
    if st is None:
        st = _arenas[idx] = {"eager": None, "cap_id": 0, "cap": None}
    cid = ctypes.c_uint64(0)
    _lib.check(lib.kpgnn_stream_capture_id(torch.cuda.current_stream(dev).cuda_stream, ctypes.byref(cid)), "kpgnn_stream_capture_id")
    if cid.value == 0:
        if st["eager"] is None:
            st["eager"] = _Arena(_EAGER_DOUBLES, dev)
        a = st["eager"]
    else:
        if st["cap_id"] != cid.value or st["cap"] is None:
            st["cap"], st["cap_id"] = _Arena(_GRAPH_DOUBLES, dev), cid.value
        a = st["cap"]
    n = (STAT_REPLICAS * 2 * C + 31) // 32 * 32
    if a.off + n > a.buf.numel():
        a.buf.zero_()
        a.off = 0
    v = a.buf[a.off:a.off + n]
    a.off += n
    return v


_COLSTATS = "_kpgnn_colstats"


def attach_column_stats(t, slot):
    """Mark `t` as carrying its column statistics (sum, sum of squares) in `slot`: a following batch_norm_act skips its
    stats pass.  Valid for this tensor object at this version only."""
    try:
        setattr(t, _COLSTATS, (t._version, slot))
    except Exception:  # pragma: no cover
        pass


def _column_stats_of(t):
    rec = getattr(t, _COLSTATS, None)
    if rec is not None and rec[0] == t._version:
        return rec[1]
    return None


class BatchNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, eps, momentum, relu, nbt=None, ready_slot=None):
        lib = _lib.load()
        x = x if x.stride(-1) == 1 else x.contiguous()
        N, C = x.shape
        dev = x.device
        slot = ready_slot if ready_slot is not None else take_stat_slot(C, dev)
        z = torch.empty((N, C), dtype=torch.float32, device=dev)
        stats = torch.empty((2, C), dtype=torch.float32, device=dev)
        d = _lib.BnDesc()
        d.N, d.C, d.relu, d.eps, d.momentum = N, C, 1 if relu else 0, eps, momentum
        d.n_dyn = dyn_ptr(N)
        d.x, d.x_stride = x.data_ptr(), x.stride(0)
        d.gamma, d.beta = gamma.data_ptr(), beta.data_ptr()
        d.running_mean, d.running_var = _ptr(running_mean), _ptr(running_var)
        d.mean, d.invstd = stats[0].data_ptr(), stats[1].data_ptr()
        d.z, d.z_stride = z.data_ptr(), z.stride(0)
        if residual is not None:
            residual = residual if residual.stride(-1) == 1 else residual.contiguous()
            d.residual, d.r_stride = residual.data_ptr(), residual.stride(0)
        d.stat_slot, d.stats_ready = slot.data_ptr(), 1 if ready_slot is not None else 0
        d.num_batches_tracked = _ptr(nbt)
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_bn_fwd(ctypes.byref(d), _stream(x)), "kpgnn_bn_fwd")
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.relu = relu
        ctx.has_res = residual is not None
        ctx.res_cell = getattr(residual, "_kp_slot_cell", None) if residual is not None else None
        return z

    @staticmethod
    def backward(ctx, dz):
        x, gamma, beta, stats = ctx.saved_tensors
        lib = _lib.load()
        dz = dz if dz.stride(-1) == 1 else dz.contiguous()
        N, C = x.shape
        dev = x.device
        slot = take_stat_slot(C, dev)
        # residual branch: when the residual is a state whose gradient is being collected in a cell (ops.state_cell: a later
        # reader already parked its share there), d/dresidual = dz is added to that buffer by the apply pass itself
        rbuf = ctx.res_cell.buf if (ctx.res_cell is not None and ctx.needs_input_grad[3]) else None
        dx = torch.empty((N, C), dtype=torch.float32, device=dev)
        dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
        d = _lib.BnBwdDesc()
        d.N, d.C, d.relu = N, C, 1 if ctx.relu else 0
        d.n_dyn = dyn_ptr(N)
        d.x, d.x_stride, d.dz, d.dz_stride = x.data_ptr(), x.stride(0), dz.data_ptr(), dz.stride(0)
        d.gamma, d.beta, d.mean, d.invstd = gamma.data_ptr(), beta.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr()
        d.dx, d.dx_stride = dx.data_ptr(), dx.stride(0)
        d.dgamma, d.dbeta = dgb[0].data_ptr(), dgb[1].data_ptr()
        d.stat_slot = slot.data_ptr()
        if rbuf is not None:
            d.residual_grad, d.rg_stride = rbuf.data_ptr(), rbuf.stride(0)
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_bn_bwd(ctypes.byref(d), _stream(x)), "kpgnn_bn_bwd")
        gres = None if (not ctx.has_res or rbuf is not None) else dz
        return dx, dgb[0], dgb[1], gres, None, None, None, None, None, None, None


def set_dense_math(mode):
    """"auto": the bf16-split products on the bf16 matrix cores where a kernel has them (include/kpgnn.h KPGNN_MATH_AUTO);
    "f32": the fp32 matrix instruction everywhere.  Applies to descriptors created afterwards."""
    _lib.DENSE_MATH = {"auto": _lib.MATH_AUTO, "f32": _lib.MATH_F32}[mode]


_LIN_WIDTHS = (32, 64, 96, 104, 128)   # lin_fused.h: fully unrolled k-loops

# kpgnn_linear_fwd: y = x W^T + b and dx = dy W for tall-skinny x on the fp32 matrix cores.  Measured 21.8 us per
# [47k,104] x [104,104] launch against 29 us for the BLAS library's kernel (profiles/r01): on by default for the shapes
# it covers (I in {32, 64, 104, 128}, O % 4 == 0, contiguous operands, N >= 1024; O > 128 walks the outputs in
# chunks of 128 over an LDS-resident x tile: 118 us vs the library's 146 us for [47k,104] x [104,936]).  Other shapes go
# to the BLAS library.  The weight-gradient kernel (34 us vs the library's 139 us) serves every shape up to 256 x 256.


def _mfma_linear(x, w, bias, transposed=False):
    """y = x w^T + bias on kpgnn_linear_fwd (w: [O,I] contiguous), or y = x w with transposed=True (w: [I,O]).
    Returns None when the shape is not covered."""
    lib = _lib.load()
    N, I = x.shape
    O = w.shape[1] if transposed else w.shape[0]
    if (O % 4 != 0 or I not in _LIN_WIDTHS or (O > 128 and I == 96) or O > 4096 or N < 1024 or not x.is_contiguous()
            or not w.is_contiguous() or x.data_ptr() % 16 or (bias is not None and bias.data_ptr() % 16)):
        return None
    y = torch.empty((N, O), dtype=torch.float32, device=x.device)
    d = _lib.LinearDesc()
    d.N, d.O, d.I = N, O, I
    d.n_dyn = dyn_ptr(N)
    d.w_transposed = 1 if transposed else 0
    d.x, d.x_stride, d.w, d.bias, d.y, d.y_stride = x.data_ptr(), x.stride(0), w.data_ptr(), _ptr(bias), y.data_ptr(), y.stride(0)
    with torch.cuda.device(x.device):
        rc = lib.kpgnn_linear_fwd(ctypes.byref(d), _stream(x))
    if rc == -3:
        return None
    _lib.check(rc, "kpgnn_linear_fwd")
    return y


class LinearWgrad(torch.autograd.Function):
    """y = x W^T + b with the GEMMs y and dx on the BLAS library and (dW, db) on the fp32-MFMA streaming
    kernel kpgnn_linear_wgrad (the library's choice for that K = N reduction runs at ~7 TFLOP/s)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        xc = x if x.stride(-1) == 1 else x.contiguous()
        y = _mfma_linear(xc, weight.contiguous(), bias)
        if y is None:
            refuse_dynamic_rows("nn.Linear outside the MFMA kernels' shapes", x.shape[0])
        return y if y is not None else F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        lib = _lib.load()
        dy = dy if dy.stride(-1) == 1 else dy.contiguous()
        x = x if x.stride(-1) == 1 else x.contiguous()
        N, O = dy.shape
        I = x.shape[1]
        dev = dy.device
        dx = None
        if ctx.needs_input_grad[0]:
            # dx = dy W: the MFMA kernel reads W in its [O,I] layout (w_transposed), no copy
            dx = _mfma_linear(dy if dy.is_contiguous() else dy.contiguous(), weight.contiguous(), None, transposed=True)
            if dx is None:
                dx = dy @ weight
        dw = torch.empty((O, I), dtype=torch.float32, device=dev)
        db = torch.empty((O,), dtype=torch.float32, device=dev) if ctx.has_bias else None
        nb = lib.kpgnn_wgrad_workspace_bytes(O, I)
        ws = torch.empty(int(nb), dtype=torch.uint8, device=dev)
        d = _lib.WgradDesc()
        d.N, d.O, d.I = N, O, I
        d.n_dyn = dyn_ptr(N)
        d.dy, d.dy_stride, d.x, d.x_stride = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0)
        d.dw, d.db, d.workspace, d.workspace_bytes = dw.data_ptr(), _ptr(db), ws.data_ptr(), int(nb)
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_linear_wgrad(ctypes.byref(d), _stream(dy)), "kpgnn_linear_wgrad")
        return dx, dw, db


def linear(x, lin):
    """nn.Linear `lin` on x [N,I]; tall-skinny fp32 CUDA inputs take the MFMA weight-gradient path."""
    if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and lin.weight.shape[0] <= 256
            and lin.weight.shape[1] <= 256 and x.shape[0] >= 1024 and torch.is_grad_enabled()
            and lin.weight.requires_grad):
        return LinearWgrad.apply(x, lin.weight, lin.bias)
    refuse_dynamic_rows("nn.Linear on the framework path", x.shape[0])
    return lin(x)


def batch_norm_act(x, bn, relu=False, residual=None):
    """nn.BatchNorm1d `bn` applied to x [N,C] (+ReLU) (+residual).  Training mode with batch statistics runs
    on the HIP kernels; everything else (eval, no affine, cumulative momentum, C > 256) on torch's GPU op."""
    use_hip = (bn.training and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and bn.affine
               and bn.momentum is not None and x.shape[1] <= 256 and x.shape[0] >= 1)
    if use_hip:
        rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
        nbt = bn.num_batches_tracked if bn.track_running_stats else None   # incremented inside the stats kernel
        return BatchNormAct.apply(x, bn.weight, bn.bias, residual, rm, rv, float(bn.eps), float(bn.momentum), relu, nbt,
                                  _column_stats_of(x))
    refuse_dynamic_rows("BatchNorm on the framework path", x.shape[0])
    out = bn(x)
    if relu:
        out = F.relu(out)
    if residual is not None:
        out = out + residual
    return out


# Split copies of the MLPs' weights for the bf16-split Linear kernels (kpgnn_linear_split_many): a body prepares ALL of them,
# both orientations, with one launch at the start of its forward; kpgnn_linear_bn then finds them here instead of splitting per
# call.  An entry serves the forward and the backward of the step that made it: the next forward overwrites it, and it is only
# taken while the weight tensor has the version it was made from.
_split_cache = {}


def invalidate_splits():
    """Forget the prepared split copies (an optimiser that updates weights without bumping their version calls this)."""
    _split_cache.clear()


def prepare_mlp_splits(mlps, num_rows):
    """One launch for every Linear of the given Linear-BatchNorm-ReLU-Linear-BatchNorm-ReLU MLPs, in the orientation of the
    forward (y = x W^T) and of the input gradient (dx = dy W).  No-op outside the bf16-split kernels' range."""
    if num_rows < 4096 or _lib.DENSE_MATH == _lib.MATH_F32:
        return
    lib = _lib.load()
    todo = []
    for mlp in mlps:
        for lin in (mlp[0], mlp[3]):
            w = lin.weight
            O, I = w.shape
            if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and O <= 104 and I <= 104 and O % 4 == 0 and I % 4 == 0):
                continue
            todo.append((w, 0, I, 1, O, I))        # forward: element (k, n) = w[n * I + k]
            todo.append((w, 1, 1, I, I, O))        # dx = dy w: output column n = input feature, k = output feature: w[k * I + n]
    if not todo:
        return
    dev = todo[0][0].device
    sizes = [int(lib.kpgnn_linear_split_workspace_bytes(O, I, 1)) for (_, _, _, _, O, I) in todo]
    buf = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
    jobs = (_lib.SplitJob * len(todo))()
    off = 0
    for j, ((w, tr, wn, wk, O, I), nb) in enumerate(zip(todo, sizes)):
        frag = buf[off:off + nb]
        jobs[j].w, jobs[j].wn, jobs[j].wk, jobs[j].O, jobs[j].I, jobs[j].frag = w.data_ptr(), wn, wk, O, I, frag.data_ptr()
        _split_cache[(w.data_ptr(), tr)] = (w._version, O, I, frag)
        off += nb
    with torch.cuda.device(dev):
        for a in range(0, len(todo), 64):
            n = min(64, len(todo) - a)
            chunk = (_lib.SplitJob * n)(*jobs[a:a + n])
            _lib.check(lib.kpgnn_linear_split_many(chunk, n, torch.cuda.current_stream(dev).cuda_stream), "kpgnn_linear_split_many")


def _lin_bn(lib, dev, **kw):
    d = _lib.LinearBnDesc()
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if torch.is_tensor(v) else v)
    d.n_dyn = dyn_ptr(kw["N"])
    w = kw["w"]
    hit = _split_cache.get((w.data_ptr(), 1 if kw.get("w_transposed", 0) else 0))
    if hit is not None and hit[0] == w._version and hit[1] == kw["O"] and hit[2] == kw["I"]:
        d.workspace, d.workspace_bytes, d.w_split_ready = hit[3].data_ptr(), hit[3].numel(), 1
    else:
        ws = _split_workspace(lib, kw["O"], kw["I"], 1, dev)          # (the bf16-split kernel's copy of W; None: fp32 kernel)
        if ws is not None:
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    with torch.cuda.device(dev):
        _lib.check(lib.kpgnn_linear_bn(ctypes.byref(d), torch.cuda.current_stream(dev).cuda_stream), "kpgnn_linear_bn")


class FusedMLP(torch.autograd.Function):
    """z = relu(bn2(relu(bn1(h W0^T + b0)) W3^T + b3)), training mode, as 3 launches forward and 5 backward
    (kpgnn_linear_bn / kpgnn_bn_fwd / kpgnn_bn_bwd / kpgnn_linear_wgrad_pair around column-statistics slots):
      fwd  y1 = h W0^T + b0 (+ column sums of y1) | y2 = relu(bn1(y1)) W3^T + b3 (bn1 applied while the tile loads; + sums
           of y2) | z = relu(bn2(y2)) (+ sums of z for the caller's next BatchNorm, when asked)
      bwd  bn2 reduce | dy2 = bn2'(dz) on load, da1 = dy2 W3 masked by relu1, bn1 reduce in the epilogue | dy1 = bn1'(da1)
           on load, dh = dy1 W0 | both weight gradients in one launch (relu(bn1(y1)) recomputed on load) + one reduce.
    The activations a1 = relu(bn1(y1)) are never written to memory.

    With an OUTER BatchNorm (gO, beO, bnO: the bodies' per-layer norm, models/GNNs.py:440-441, + optional residual) the
    node returns bnO(z) + residual: one more apply launch forward, and backward STILL 5 launches - the outer norm's reduce,
    its apply and bn2's reduce (three passes, dz written and read back) become one stacked reduce over (dh, y2), and both
    norms' backward arithmetic happens while the first GEMM loads its tile (kpgnn_linear_bn pro 3)."""

    @staticmethod
    def forward(ctx, h, w0, b0, g1, be1, w3, b3, g2, be2, bn1, bn2, out_slot, gO=None, beO=None, residual=None, bnO=None):
        lib = _lib.load()
        dev = h.device
        N, I = h.shape
        O = w0.shape[0]
        h = h.contiguous()
        w0c, w3c = w0.contiguous(), w3.contiguous()
        outer = bnO is not None
        slot1, slot2 = take_stat_slot(O, dev), take_stat_slot(O, dev)
        if outer:
            out_slot = take_stat_slot(O, dev)
        y1 = torch.empty((N, O), dtype=torch.float32, device=dev)
        y2 = torch.empty((N, O), dtype=torch.float32, device=dev)
        st = torch.empty((6, O), dtype=torch.float32, device=dev)      # mean1, invstd1, mean2, invstd2, meanO, invstdO
        _lin_bn(lib, dev, N=N, O=O, I=I, x=h, w=w0c, bias=b0, y=y1, pro=0, epi=1, out_slot=slot1)
        track1 = bn1.track_running_stats
        _lin_bn(lib, dev, N=N, O=O, I=O, x=y1, w=w3c, bias=b3, y=y2, pro=1, epi=1, pro_relu=1, in_slot=slot1, in_gamma=g1,
                in_beta=be1, in_eps=float(bn1.eps), momentum=float(bn1.momentum), in_mean=st[0], in_invstd=st[1],
                running_mean=bn1.running_mean if track1 else None, running_var=bn1.running_var if track1 else None,
                num_batches_tracked=bn1.num_batches_tracked if track1 else None, out_slot=slot2)
        d = _lib.BnDesc()
        d.N, d.C, d.relu, d.eps, d.momentum = N, O, 1, float(bn2.eps), float(bn2.momentum)
        d.n_dyn = dyn_ptr(N)
        d.x, d.x_stride, d.gamma, d.beta = y2.data_ptr(), O, g2.data_ptr(), be2.data_ptr()
        if bn2.track_running_stats:
            d.running_mean, d.running_var = bn2.running_mean.data_ptr(), bn2.running_var.data_ptr()
            d.num_batches_tracked = bn2.num_batches_tracked.data_ptr()
        out = torch.empty((N, O), dtype=torch.float32, device=dev)
        d.mean, d.invstd, d.z, d.z_stride = st[2].data_ptr(), st[3].data_ptr(), out.data_ptr(), O
        d.stat_slot, d.stats_ready, d.out_slot = slot2.data_ptr(), 1, _ptr(out_slot)
        ctx.outer, ctx.has_res, ctx.res_cell = outer, False, None
        if outer:
            # the body's norm (+ residual) by the same call: a statistics-only pass over y2, then one apply pass for both norms;
            # z = relu(bn2(y2)) is never written (backward recomputes it from y2 as well)
            d.outer_gamma, d.outer_beta = gO.data_ptr(), beO.data_ptr()
            d.outer_eps, d.outer_momentum = float(bnO.eps), float(bnO.momentum)
            if bnO.track_running_stats:
                d.outer_running_mean, d.outer_running_var = bnO.running_mean.data_ptr(), bnO.running_var.data_ptr()
                d.outer_num_batches_tracked = bnO.num_batches_tracked.data_ptr()
            d.outer_mean, d.outer_invstd = st[4].data_ptr(), st[5].data_ptr()
            if residual is not None:
                residual = residual if residual.stride(-1) == 1 else residual.contiguous()
                d.residual, d.r_stride = residual.data_ptr(), residual.stride(0)
                ctx.has_res = True
                ctx.res_cell = getattr(residual, "_kp_slot_cell", None)
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_bn_fwd(ctypes.byref(d), _stream(h)), "kpgnn_bn_fwd")
        if outer:
            ctx.save_for_backward(h, w0c, w3c, g1, be1, g2, be2, y1, y2, st, gO)
        else:
            ctx.save_for_backward(h, w0c, w3c, g1, be1, g2, be2, y1, y2, st)
        ctx.has_b0, ctx.has_b3 = b0 is not None, b3 is not None
        return out

    @staticmethod
    def backward(ctx, dz):
        if ctx.outer:
            h, w0, w3, g1, be1, g2, be2, y1, y2, st, gO = ctx.saved_tensors
        else:
            h, w0, w3, g1, be1, g2, be2, y1, y2, st = ctx.saved_tensors
        lib = _lib.load()
        dev = h.device
        N, I = h.shape
        O = w0.shape[0]
        dz = dz.contiguous()
        slot2, slot1 = take_stat_slot(4 * O if ctx.outer else O, dev), take_stat_slot(O, dev)
        dy2 = torch.empty((N, O), dtype=torch.float32, device=dev)
        da1 = torch.empty((N, O), dtype=torch.float32, device=dev)
        dy1 = torch.empty((N, O), dtype=torch.float32, device=dev)
        dh = torch.empty((N, I), dtype=torch.float32, device=dev)
        gb = torch.empty((6, O), dtype=torch.float32, device=dev)      # dgamma2, dbeta2, dgamma1, dbeta1, dgammaO, dbetaO
        d = _lib.BnBwdDesc()
        d.N, d.C, d.relu = N, O, 1
        d.n_dyn = dyn_ptr(N)
        d.x, d.x_stride, d.dz, d.dz_stride = y2.data_ptr(), O, dz.data_ptr(), O
        d.gamma, d.beta, d.mean, d.invstd = g2.data_ptr(), be2.data_ptr(), st[2].data_ptr(), st[3].data_ptr()
        d.stat_slot, d.reduce_only = slot2.data_ptr(), 1
        rbuf = None
        handed = False
        if ctx.outer:
            from . import ops
            # the residual branch: d/dresidual is dz itself.  A state whose gradient is collected in a cell gets it either as one
            # more addend of its pull gather (no pass at all) or += from the reduce pass below
            want_res = ctx.res_cell is not None and ctx.needs_input_grad[14]
            if want_res and ops.pull_applies(N, O) and ctx.res_cell.pull_reader:
                ctx.res_cell.park_addend(dz)
                handed = True
            elif want_res:
                rbuf = ctx.res_cell.buf
            d.outer_mean, d.outer_invstd = st[4].data_ptr(), st[5].data_ptr()
            if rbuf is not None:
                d.residual_grad, d.rg_stride = rbuf.data_ptr(), rbuf.stride(0)
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_bn_bwd(ctypes.byref(d), _stream(h)), "kpgnn_bn_bwd")
        okw = dict(pro=3, o_mean=st[4], o_invstd=st[5], o_gamma=gO, o_dgamma=gb[4], o_dbeta=gb[5]) if ctx.outer else dict(pro=2)
        _lin_bn(lib, dev, N=N, O=O, I=O, x=dz, w=w3, y=da1, w_transposed=1, epi=2, pro_relu=1, in_slot=slot2,
                in_gamma=g2, in_beta=be2, in_mean=st[2], in_invstd=st[3], x2=y2, xt=dy2, dgamma=gb[0], dbeta=gb[1],
                out_slot=slot1, e_x=y1, e_mean=st[0], e_invstd=st[1], e_gamma=g1, e_beta=be1, **okw)
        _lin_bn(lib, dev, N=N, O=I, I=O, x=da1, w=w0, y=dh, w_transposed=1, pro=2, epi=0, pro_relu=0, in_slot=slot1,
                in_gamma=g1, in_beta=be1, in_mean=st[0], in_invstd=st[1], x2=y1, xt=dy1, dgamma=gb[2], dbeta=gb[3])
        # weight gradients: dW3 = dy2^T relu(bn1(y1)), dW0 = dy1^T h
        dw3 = torch.empty((O, O), dtype=torch.float32, device=dev)
        dw0 = torch.empty((O, I), dtype=torch.float32, device=dev)
        db = torch.empty((2, O), dtype=torch.float32, device=dev)
        a, b = _lib.WgradDesc(), _lib.WgradDesc()
        a.N, a.O, a.I = N, O, O
        a.n_dyn = b.n_dyn = dyn_ptr(N)
        a.dy, a.dy_stride, a.x, a.x_stride = dy2.data_ptr(), O, y1.data_ptr(), O
        a.dw, a.db = dw3.data_ptr(), db[0].data_ptr()
        a.x_mean, a.x_invstd, a.x_gamma, a.x_beta, a.x_relu = st[0].data_ptr(), st[1].data_ptr(), g1.data_ptr(), be1.data_ptr(), 1
        b.N, b.O, b.I = N, O, I
        b.dy, b.dy_stride, b.x, b.x_stride = dy1.data_ptr(), O, h.data_ptr(), I
        b.dw, b.db = dw0.data_ptr(), db[1].data_ptr()
        with torch.cuda.device(dev):
            if I == O:
                from . import ops
                nb = 2 * int(lib.kpgnn_wgrad_workspace_bytes(O, O))
                ws = torch.empty(nb, dtype=torch.uint8, device=dev)
                a.workspace, a.workspace_bytes = ws.data_ptr(), nb
                job = ops.defer_reduce_job(w0, w3)    # (inside ops.deferred_reductions(): the reduce rides with a later launch)
                if job is not None:
                    a.defer = ctypes.cast(ctypes.pointer(job), ctypes.c_void_p)
                _lib.check(lib.kpgnn_linear_wgrad_pair(ctypes.byref(a), ctypes.byref(b), _stream(h)), "kpgnn_linear_wgrad_pair")
                if job is not None:
                    ops.queue_reduce_job(job, (ws, dw3, dw0, db))
            else:
                for q in (a, b):
                    nb = int(lib.kpgnn_wgrad_workspace_bytes(q.O, q.I))
                    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
                    q.workspace, q.workspace_bytes = ws.data_ptr(), nb
                    _lib.check(lib.kpgnn_linear_wgrad(ctypes.byref(q), _stream(h)), "kpgnn_linear_wgrad")
        gres = dz if (ctx.outer and ctx.has_res and rbuf is None and not handed) else None
        return (dh if ctx.needs_input_grad[0] else None, dw0, db[1] if ctx.has_b0 else None, gb[2], gb[3],
                dw3, db[0] if ctx.has_b3 else None, gb[0], gb[1], None, None, None,
                gb[4] if ctx.outer else None, gb[5] if ctx.outer else None, gres, None)


def _fusable_bn(bn):
    return bn.training and bn.affine and bn.momentum is not None


def mlp_linear_bn_relu_x2(mlp, h, emit_out_stats=False, post_norm=None):
    """nn.Sequential(Linear, BatchNorm1d, ReLU, Linear, BatchNorm1d, ReLU) (KPGINplus.py:25-30, gine.py:31-38).
    Training mode on covered widths: the fused 3 + 5 launch path (FusedMLP); otherwise Linear and BatchNorm one by one.
    emit_out_stats: also accumulate the column statistics of the result and attach them to it, for a BatchNorm the
    caller applies next (the bodies' per-layer norm).
    post_norm = (nn.BatchNorm1d, residual or None): the caller's next step IS bn(result) + residual - returned instead of
    the MLP's output, by the same autograd node when fused (FusedMLP: the backward saves two passes)."""
    l0, bn1, l3, bn2 = mlp[0], mlp[1], mlp[3], mlp[4]
    O, I = l0.weight.shape
    bnO, res = post_norm if post_norm is not None else (None, None)
    if (h.is_cuda and h.dim() == 2 and h.dtype == torch.float32 and torch.is_grad_enabled() and _fusable_bn(bn1)
            and _fusable_bn(bn2) and I in _LIN_WIDTHS and O in _LIN_WIDTHS and tuple(l3.weight.shape) == (O, O)
            and h.shape[0] >= 1 and h.data_ptr() % 16 == 0
            and (l0.bias is None or l0.bias.data_ptr() % 16 == 0) and (l3.bias is None or l3.bias.data_ptr() % 16 == 0)):
        if (bnO is not None and _fusable_bn(bnO) and bnO.weight.data_ptr() % 16 == 0 and bnO.bias.data_ptr() % 16 == 0
                and (res is None or (res.is_cuda and res.dtype == torch.float32 and tuple(res.shape) == (h.shape[0], O)
                                     and res.stride(0) % 4 == 0 and res.data_ptr() % 16 == 0))):
            return FusedMLP.apply(h, l0.weight, l0.bias, bn1.weight, bn1.bias, l3.weight, l3.bias, bn2.weight, bn2.bias,
                                  bn1, bn2, None, bnO.weight, bnO.bias, res, bnO)
        out_slot = take_stat_slot(O, h.device) if (emit_out_stats or bnO is not None) else None
        z = FusedMLP.apply(h, l0.weight, l0.bias, bn1.weight, bn1.bias, l3.weight, l3.bias, bn2.weight, bn2.bias, bn1, bn2,
                           out_slot)
        if out_slot is not None:
            attach_column_stats(z, out_slot)
    else:
        z = batch_norm_act(linear(h, mlp[0]), mlp[1], relu=True)
        z = batch_norm_act(linear(z, mlp[3]), mlp[4], relu=True)
    return z if bnO is None else batch_norm_act(z, bnO, relu=False, residual=res)


# ------------------------------------------------------------------------------------ jumping-knowledge projection
def _jk_native_ok(weight, bias, states):
    """Shapes the grouped-K kernels take: S <= 16 contiguous fp32 [N,H] states of one width H in {32, 64, 96, 104, 128},
    O <= 128 with O % 4 == 0, S * H > 128 (the blocked input-gradient kernel), 16-B aligned operands."""
    S = len(states)
    N, H = states[0].shape
    O = weight.shape[0]
    return (2 <= S <= 16 and H in _LIN_WIDTHS and O % 4 == 0 and O in (32, 64, 104, 128) and S * H > 128 and N >= 1
            and tuple(weight.shape) == (O, S * H) and weight.is_contiguous() and weight.data_ptr() % 16 == 0
            and (bias is None or (bias.is_contiguous() and bias.data_ptr() % 16 == 0))
            and all(st.shape == (N, H) and st.dtype == torch.float32 and st.is_contiguous() and st.data_ptr() % 16 == 0 for st in states))


def _split_workspace(lib, O, I, group, device):
    """Scratch for the bf16-split Linear kernels (kpgnn_linear_split_workspace_bytes): the copy of the weight in matrix-fragment
    order, rebuilt by every call (the weights change every step); None when the kernels do not take the shape."""
    nb = int(lib.kpgnn_linear_split_workspace_bytes(O, I, group))
    return torch.empty(nb, dtype=torch.uint8, device=device) if nb > 0 else None


class JKConcatLinear(torch.autograd.Function):
    """relu(cat(states, dim=1) W^T + b): the bodies' jumping-knowledge projection (models/GNNs.py:216-218, output_proj).
    Native path (kpgnn_linear_group_fwd / kpgnn_linear_fwd / kpgnn_linear_wgrad_group): the concatenation is never made - the
    GEMM's K-loop runs over the state pointers; backward reads the ReLU mask from the saved output while it loads dy
    (no masked copy), the input gradient of all S states comes out of ONE launch as S contiguous [N,H] matrices, and the
    weight gradient's S column blocks run side by side in one launch.  A state whose gradient is being collected in a cell
    (ops.state_cell: GNNPlus' hop-slot history) gets its matrix PARKED there - the layers' gather kernels and the norms'
    residual branches then add into it in place and the state's last reader hands autograd the total, instead of autograd
    summing S + 2 strided tensors per state (36 adds of ~20 MB per step at K = L = 8).
    Shapes outside the kernels' limits keep the reference's concat + library GEMM."""

    @staticmethod
    def forward(ctx, weight, bias, *states):
        ctx.native = _jk_native_ok(weight, bias, states)
        ctx.cells = [getattr(st, "_kp_slot_cell", None) for st in states]
        ctx.widths = [st.shape[1] for st in states]
        ctx.has_bias = bias is not None
        if ctx.native:
            lib = _lib.load()
            N, H = states[0].shape
            O = weight.shape[0]
            y = torch.empty((N, O), dtype=torch.float32, device=weight.device)
            d = _lib.LinearGroupDesc()
            d.N, d.O, d.I, d.group = N, O, H, len(states)
            d.n_dyn = dyn_ptr(N)
            for l, st in enumerate(states):
                d.x[l] = st.data_ptr()
            d.x_stride, d.w, d.bias, d.y, d.relu = H, weight.data_ptr(), _ptr(bias), y.data_ptr(), 1
            ws = _split_workspace(lib, O, H, len(states), weight.device)     # (the bf16-split kernel's copy of W; None: fp32 kernel)
            if ws is not None:
                d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
            with torch.cuda.device(weight.device):
                _lib.check(lib.kpgnn_linear_group_fwd(ctypes.byref(d), _stream(y)), "kpgnn_linear_group_fwd")
            ctx.save_for_backward(weight, y, *states)
            return y
        refuse_dynamic_rows("the jumping-knowledge projection outside the grouped-K kernels' shapes", states[0].shape[0])
        rep = torch.cat(states, dim=1)
        if bias is not None and hasattr(torch, "_addmm_activation"):
            # the library GEMM with the bias + ReLU epilogue fused (bitwise the same as addmm + relu_; 117 vs 128 us at [47k, 936])
            y = torch._addmm_activation(bias, rep, weight.t())
        else:
            y = torch.addmm(bias, rep, weight.t()) if bias is not None else rep @ weight.t()
            y.relu_()
        ctx.save_for_backward(weight, rep, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        S, H = len(ctx.widths), ctx.widths[0]
        lib = _lib.load()
        dy = dy.contiguous()
        if ctx.native and dy.data_ptr() % 16 == 0:
            weight, y, *states = ctx.saved_tensors
            N, O = y.shape
            dev = dy.device
            dw = db = None
            with torch.cuda.device(dev):
                G = torch.empty((S, N, H), dtype=torch.float32, device=dev)
                d = _lib.LinearDesc()
                d.N, d.O, d.I = N, S * H, O
                d.n_dyn = dyn_ptr(N)
                d.x, d.x_stride, d.w, d.y, d.y_stride = dy.data_ptr(), O, weight.data_ptr(), G.data_ptr(), H
                d.w_transposed, d.y_block_cols, d.y_block_stride = 1, H, N * H
                d.x_mask = y.data_ptr()                       # dL/d(pre-activation) = dy where the saved output is > 0
                wsx = _split_workspace(lib, H, O, S, dev)
                if wsx is not None:
                    d.workspace, d.workspace_bytes = wsx.data_ptr(), wsx.numel()
                _lib.check(lib.kpgnn_linear_fwd(ctypes.byref(d), _stream(dy)), "kpgnn_linear_fwd")
                if ctx.needs_input_grad[0] or (ctx.has_bias and ctx.needs_input_grad[1]):
                    from . import ops
                    dw = torch.empty((O, S * H), dtype=torch.float32, device=dev)
                    db = torch.empty((O,), dtype=torch.float32, device=dev)
                    nb = int(lib.kpgnn_wgrad_group_workspace_bytes(O, H, S))
                    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
                    q = _lib.WgradDesc()
                    q.N, q.O, q.I = N, O, H
                    q.n_dyn = dyn_ptr(N)
                    q.dy, q.dy_stride, q.x, q.x_stride = dy.data_ptr(), O, states[0].data_ptr(), H
                    q.dy_mask = y.data_ptr()
                    q.dw, q.db, q.workspace, q.workspace_bytes = dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb
                    xs = (ctypes.c_void_p * S)(*[st.data_ptr() for st in states])
                    job = ops.defer_reduce_job(weight)     # (inside ops.deferred_reductions(): the reduce rides with a later launch)
                    if job is not None:
                        q.defer = ctypes.cast(ctypes.pointer(job), ctypes.c_void_p)
                    _lib.check(lib.kpgnn_linear_wgrad_group(ctypes.byref(q), xs, S, _stream(dy)), "kpgnn_linear_wgrad_group")
                    if job is not None:
                        ops.queue_reduce_job(job, (ws, dw, db))
            parts = [G[l] for l in range(S)]
            if not ctx.has_bias:
                db = None
        else:
            if ctx.native:
                weight, y, *states = ctx.saved_tensors
                rep = torch.cat(states, dim=1)
            else:
                weight, rep, y = ctx.saved_tensors
            N, O = y.shape
            dym = torch.ops.aten.threshold_backward(dy, y, 0.0)
            dw = dym.t() @ rep if ctx.needs_input_grad[0] else None
            db = dym.sum(0) if (ctx.has_bias and ctx.needs_input_grad[1]) else None
            G = dym @ weight
            parts, c0 = [], 0
            for w in ctx.widths:
                parts.append(G[:, c0:c0 + w])
                c0 += w
        grads = []
        for l, cell in enumerate(ctx.cells):
            if not ctx.needs_input_grad[2 + l]:
                grads.append(None)
            elif cell is not None and cell.buf is None and parts[l].is_contiguous():
                cell.buf = parts[l]                       # parked: the state's remaining readers add to it in place
                grads.append(None)
            else:
                grads.append(parts[l])
        return (dw, db, *grads)


# ------------------------------------------------------------------- KP-GIN per-hop MLP (+ geometric combine + projection)
class HopMlp(torch.autograd.Function):
    """relu(relu(s W1 + b1) W2 + b2) per hop (+ sum_k theta_k * . (+ combine_proj)), one HIP launch per direction
    (hop_mlp.hip)."""

    @staticmethod
    def forward(ctx, s, w1, b1, w2, b2, theta, wc, bc):
        lib = _lib.load()
        s = s.contiguous()
        N, K, DI = s.shape
        refuse_dynamic_rows("kpgnn_hop_mlp", N)
        DO = w1.shape[2]
        H = wc.shape[0] if wc is not None else 0
        dev = s.device
        w1, b1, w2, b2 = w1.contiguous(), b1.contiguous(), w2.contiguous(), b2.contiguous()
        theta = theta.contiguous() if theta is not None else None
        wc = wc.contiguous() if wc is not None else None
        bc = bc.contiguous() if bc is not None else None
        h = torch.empty((2, N, K, DO), dtype=torch.float32, device=dev)
        out = torch.empty((N, H if H else DO), dtype=torch.float32, device=dev) if theta is not None else None
        d = _lib.HopMlpDesc()
        d.N, d.K, d.DI, d.DO, d.H = N, K, DI, DO, H
        d.s, d.w1, d.b1, d.w2, d.b2, d.theta = s.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), _ptr(theta)
        d.wc, d.bc = _ptr(wc), _ptr(bc)
        d.h1, d.h2, d.out = h[0].data_ptr(), h[1].data_ptr(), _ptr(out)
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_hop_mlp_fwd(ctypes.byref(d), _stream(s)), "kpgnn_hop_mlp_fwd")
        ctx.save_for_backward(s, w1, b1, w2, b2, theta, wc, bc, h)
        return out if theta is not None else h[1]

    @staticmethod
    def backward(ctx, gout):
        s, w1, b1, w2, b2, theta, wc, bc, h = ctx.saved_tensors
        lib = _lib.load()
        N, K, DI = s.shape
        DO = w1.shape[2]
        H = wc.shape[0] if wc is not None else 0
        dev = s.device
        gout = gout.contiguous()
        gs = torch.empty((N, K, DI), dtype=torch.float32, device=dev)
        sizes = [K * DI * DO, K * DO, K * DO * DO, K * DO] + ([K * DO] if theta is not None else []) + ([H * DO, H] if H else [])
        gflat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        ws_bytes = int(lib.kpgnn_hop_mlp_workspace_bytes(max(N, 1), K, DI, DO, H))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        d = _lib.HopMlpDesc()
        d.N, d.K, d.DI, d.DO, d.H = N, K, DI, DO, H
        d.s, d.w1, d.b1, d.w2, d.b2, d.theta = s.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), _ptr(theta)
        d.wc, d.bc = _ptr(wc), _ptr(bc)
        d.h1, d.h2 = h[0].data_ptr(), h[1].data_ptr()
        d.gout, d.gs, d.gflat = gout.data_ptr(), gs.data_ptr(), gflat.data_ptr()
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws_bytes
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_hop_mlp_bwd(ctypes.byref(d), _stream(s)), "kpgnn_hop_mlp_bwd")
        parts = list(torch.split(gflat, sizes))
        gth = parts[4].view(K, DO) if theta is not None else None
        gwc = parts[-2].view(H, DO) if H else None
        gbc = parts[-1] if (H and bc is not None) else None
        return gs, parts[0].view(K, DI, DO), parts[1].view(K, DO), parts[2].view(K, DO, DO), parts[3].view(K, DO), gth, gwc, gbc


def hop_mlp_supported(K, DI, DO, H=0):
    """Whether kpgnn_hop_mlp_* covers the shape (wider hops keep the batched-matmul path)."""
    return int(_lib.load().kpgnn_hop_mlp_workspace_bytes(1, K, DI, DO, H)) > 0


def hop_mlp(s, w1, b1, w2, b2, theta=None, wc=None, bc=None):
    """s [N,K,DI] -> h2 [N,K,DO] (theta None), comb [N,DO] = sum_k theta[k] * h2[:,k] (theta), or
    comb wc^T + bc [N,H] (theta and wc)  (reference KPGIN.py:106-112)."""
    if not s.is_cuda:
        raise _lib.KpgnnError("hop_mlp needs device tensors: there is no CPU path")
    if wc is not None and theta is None:
        raise _lib.KpgnnError("hop_mlp: the projection needs theta")
    return HopMlp.apply(s, w1, b1, w2, b2, theta, wc, bc)


# ------------------------------------------------------------------------------------ geometric hop-combine weights
class GeoTheta(torch.autograd.Function):
    """theta[k,d] = softmax_k(a (1-a)^k), a = sigmoid(alphas[d])  (combine.py:43-50): one launch per direction."""

    @staticmethod
    def forward(ctx, alphas, K):
        lib = _lib.load()
        alphas = alphas.contiguous()
        D = alphas.numel()
        theta = torch.empty((K, D), dtype=torch.float32, device=alphas.device)
        with torch.cuda.device(alphas.device):
            _lib.check(lib.kpgnn_geo_theta_fwd(alphas.data_ptr(), K, D, theta.data_ptr(), _stream(alphas)), "kpgnn_geo_theta_fwd")
        ctx.save_for_backward(alphas, theta)
        ctx.K = K
        return theta

    @staticmethod
    def backward(ctx, gtheta):
        alphas, theta = ctx.saved_tensors
        lib = _lib.load()
        gtheta = gtheta.contiguous()
        D = alphas.numel()
        ga = torch.empty_like(alphas)
        with torch.cuda.device(alphas.device):
            _lib.check(lib.kpgnn_geo_theta_bwd(alphas.data_ptr(), theta.data_ptr(), gtheta.data_ptr(), ctx.K, D, ga.data_ptr(),
                                               _stream(alphas)), "kpgnn_geo_theta_bwd")
        return ga, None


def geo_theta(alphas, K):
    return GeoTheta.apply(alphas, K)


# ------------------------------------------------------------------------------------------- graph regressor
class ScoreHead(torch.autograd.Function):
    """nn.Linear(hidden, 1) on the pooled graph rows (reference models/GraphRegression.py:17,46-51) as one launch per pass
    (kpgnn_score_head_fwd / _bwd) instead of two library GEMM launches of ~10.7 us and a bias reduce."""

    @staticmethod
    def forward(ctx, pooled, weight, bias):
        p = pooled.contiguous()
        w = weight.reshape(-1).contiguous()
        G, D = p.shape
        score = torch.empty((G, 1), dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            _lib.check(_lib.load().kpgnn_score_head_fwd(p.data_ptr(), w.data_ptr(), _ptr(bias), G, D, score.data_ptr(), _stream(p)),
                       "kpgnn_score_head_fwd")
        ctx.save_for_backward(p, w)
        ctx.has_bias = bias is not None
        return score

    @staticmethod
    def backward(ctx, ds):
        p, w = ctx.saved_tensors
        G, D = p.shape
        ds = ds.reshape(-1).contiguous()
        dp_ = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        dw = torch.empty((1, D), dtype=torch.float32, device=p.device)
        db = torch.empty(1, dtype=torch.float32, device=p.device) if ctx.has_bias else None
        with torch.cuda.device(p.device):
            _lib.check(_lib.load().kpgnn_score_head_bwd(p.data_ptr(), w.data_ptr(), ds.data_ptr(), G, D, _ptr(dp_), dw.data_ptr(),
                                                        _ptr(db), _stream(p)), "kpgnn_score_head_bwd")
        return dp_, dw, db


def score_head(pooled, lin):
    """lin(pooled) for the regressor nn.Linear(hidden, 1); any other shape / device / dtype goes to the module itself."""
    if (lin.out_features == 1 and pooled.is_cuda and pooled.dim() == 2 and pooled.dtype == torch.float32
            and lin.weight.dtype == torch.float32 and pooled.shape[0] >= 1):
        return ScoreHead.apply(pooled, lin.weight, lin.bias)
    return lin(pooled)


# ------------------------------------------------------------------------------------------- regression loss
class RegressionLoss(torch.autograd.Function):
    """mean |score - y| (kind 0, train_ZINC.py:42) or mean (score - y)^2 (kind 1, train_qm9.py:96) with its gradient
    computed by the same launch (kpgnn_regression_loss): 1 + 1 launches instead of the framework's 3 + 4."""

    @staticmethod
    def forward(ctx, score, y, kind):
        s = score.reshape(-1).contiguous()
        t = y.reshape(-1).to(torch.float32).contiguous()
        assert s.numel() == t.numel() and s.dtype == torch.float32
        loss = torch.empty((), dtype=torch.float32, device=s.device)
        ds = torch.empty_like(s) if ctx.needs_input_grad[0] else None
        with torch.cuda.device(s.device):
            _lib.check(_lib.load().kpgnn_regression_loss(s.data_ptr(), t.data_ptr(), s.numel(), kind, loss.data_ptr(), _ptr(ds),
                                                         _stream(s)), "kpgnn_regression_loss")
        ctx.save_for_backward(ds)
        ctx.shape = score.shape
        return loss

    @staticmethod
    def backward(ctx, gout):
        (ds,) = ctx.saved_tensors
        return (ds * gout).view(ctx.shape) if ds is not None else None, None, None


def regression_loss_and_grad(score, y, kind="l1"):
    """(loss, d loss / d score) of regression_loss from its one launch, for a caller that seeds the backward pass itself
    (torch.autograd.grad(score, params, grad_outputs=dscore)): the ones-fill and the multiply by it never run."""
    with torch.no_grad():
        s = score.detach().reshape(-1).contiguous()
        t = y.reshape(-1).to(torch.float32).contiguous()
        assert s.is_cuda and s.dtype == torch.float32 and s.numel() == t.numel() and s.numel() >= 1
        loss = torch.empty((), dtype=torch.float32, device=s.device)
        ds = torch.empty_like(s)
        with torch.cuda.device(s.device):
            _lib.check(_lib.load().kpgnn_regression_loss(s.data_ptr(), t.data_ptr(), s.numel(), 0 if kind == "l1" else 1,
                                                         loss.data_ptr(), ds.data_ptr(), _stream(s)), "kpgnn_regression_loss")
    return loss, ds.view(score.shape)


def regression_loss(score, y, kind="l1"):
    """The training scripts' loss on a batch of graph scores: kind "l1" = (score.squeeze() - y.squeeze()).abs().mean()
    (train_ZINC.py:42), "mse" = its squared counterpart (train_qm9.py:96)."""
    if score.is_cuda and score.dtype == torch.float32 and score.numel() == y.numel() and score.numel() >= 1:
        return RegressionLoss.apply(score, y, 0 if kind == "l1" else 1)
    d = score.squeeze() - y.squeeze()
    return d.abs().mean() if kind == "l1" else (d * d).mean()
