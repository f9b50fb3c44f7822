"""Attention hop-combine operator (reference layers/combine.py:22-27) on the HIP kernels of attention.hip.

    score = sum_c biLSTM(x)[:, :, c];  w = softmax_k(score);  out = sum_k w[:, k] * x[:, k, :]

The [N*K, D] x [D, 8K] input projection is a library GEMM (matrix cores); the K-step recurrence, softmax,
weighted sum and the whole BPTT run in hand-written kernels; dW_ih / db / dW_hh are weight-gradient GEMMs of
the BPTT output on the fp32-MFMA streaming kernel (kpgnn_linear_wgrad)."""
import ctypes

import torch

from . import _lib
from .ops import _last_contig, _ptr, _stream, refuse_dynamic_rows


def _wgrad(dy, x, want_bias):
    """dW [O,I] = dy^T x, db [O] = sum dy over rows; dy / x are 2-D row-strided views."""
    lib = _lib.load()
    N, O = dy.shape
    I = x.shape[1]
    dev = dy.device
    dw = torch.empty((O, I), dtype=torch.float32, device=dev)
    db = torch.empty((O,), dtype=torch.float32, device=dev) if want_bias else None
    nb = lib.kpgnn_wgrad_workspace_bytes(O, I)
    ws = torch.empty(int(nb), dtype=torch.uint8, device=dev)
    d = _lib.WgradDesc()
    d.N, d.O, d.I = N, O, I
    d.dy, d.dy_stride, d.x, d.x_stride = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0)
    d.dw, d.db, d.workspace, d.workspace_bytes = dw.data_ptr(), _ptr(db), ws.data_ptr(), int(nb)
    with torch.cuda.device(dev):
        _lib.check(lib.kpgnn_linear_wgrad(ctypes.byref(d), _stream(dy)), "kpgnn_linear_wgrad")
    return dw, db


class AttentionCombineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        lib = _lib.load()
        x = _last_contig(x.float())
        N, K, D = x.shape
        refuse_dynamic_rows("the attention combine (its products run over every row)", N)
        dev = x.device
        w_cat = torch.cat([w_ih, w_ih_r], dim=0)                       # [8K, D]
        b_cat = torch.cat([b_ih + b_hh, b_ih_r + b_hh_r], dim=0)        # [8K]
        xf = x.reshape(N * K, D)                                        # (copies only if x is a strided view)
        gin = torch.addmm(b_cat, xf, w_cat.t())                         # [N*K, 8K] = [N,K,2,4K]
        whh = torch.stack([w_hh, w_hh_r], dim=0).contiguous()           # [2,4K,K]
        acts = torch.empty((2, N, K, 5 * K), dtype=torch.float32, device=dev)
        hsum = torch.empty((2, N, K), dtype=torch.float32, device=dev)
        w = torch.empty((N, K), dtype=torch.float32, device=dev)
        out = torch.empty((N, D), dtype=torch.float32, device=dev)
        d = _lib.AttnDesc()
        d.N, d.K, d.D = N, K, D
        d.x, d.x_sn, d.x_sk = x.data_ptr(), x.stride(0), x.stride(1)
        d.gin, d.whh, d.acts, d.hsum, d.w, d.out = (gin.data_ptr(), whh.data_ptr(), acts.data_ptr(), hsum.data_ptr(),
                                                    w.data_ptr(), out.data_ptr())
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_attn_fwd(ctypes.byref(d), _stream(x)), "kpgnn_attn_fwd")
        ctx.save_for_backward(x, w_cat, whh, acts, w)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, w_cat, whh, acts, w = ctx.saved_tensors
        lib = _lib.load()
        gout = gout.contiguous()
        N, K, D = x.shape
        dev = x.device
        dx = torch.empty((N, K, D), dtype=torch.float32, device=dev)
        ds = torch.empty((N, K), dtype=torch.float32, device=dev)
        dgin = torch.empty((N * K, 8 * K), dtype=torch.float32, device=dev)
        hprev = torch.empty((N * K, 2 * K), dtype=torch.float32, device=dev)
        hsum = torch.empty((1,), dtype=torch.float32, device=dev)       # unused in backward
        gin_dummy = dgin                                                # (only validated for non-NULL)
        d = _lib.AttnDesc()
        d.N, d.K, d.D = N, K, D
        d.x, d.x_sn, d.x_sk = x.data_ptr(), x.stride(0), x.stride(1)
        d.gin, d.whh, d.acts, d.hsum, d.w = gin_dummy.data_ptr(), whh.data_ptr(), acts.data_ptr(), hsum.data_ptr(), w.data_ptr()
        d.gout, d.dx, d.ds, d.dgin, d.hprev = gout.data_ptr(), dx.data_ptr(), ds.data_ptr(), dgin.data_ptr(), hprev.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_attn_bwd(ctypes.byref(d), _stream(x)), "kpgnn_attn_bwd")
        # dx += dgin @ W_ih   (library GEMM, accumulated into the direct part written by the kernel)
        dxf = dx.view(N * K, D)
        dxf.addmm_(dgin, w_cat)
        # weight gradients: [8K, D] and bias [8K] from (dgin, x); recurrent [4K, K] per direction from (dgin, hprev)
        xf = x.reshape(N * K, D)
        dw_cat, db_cat = _wgrad(dgin, xf, True)
        dwhh_f, _ = _wgrad(dgin[:, :4 * K], hprev[:, :K], False)
        dwhh_r, _ = _wgrad(dgin[:, 4 * K:], hprev[:, K:], False)
        return (dx, dw_cat[:4 * K], dwhh_f, db_cat[:4 * K], db_cat[:4 * K],
                dw_cat[4 * K:], dwhh_r, db_cat[4 * K:], db_cat[4 * K:])


SCAN = True            # the scan form of the operator (input projection inside, K <= 8) where it applies
SCAN_MIN_N = 1024      # below this the launch-count difference is all there is


def scan_applies(x, K, D):
    return (SCAN and x.is_cuda and x.dtype == torch.float32 and K <= 8 and D % 4 == 0 and D <= 128 and x.shape[0] >= SCAN_MIN_N
            and x.stride(2) == 1 and x.stride(0) % 4 == 0 and x.stride(1) % 4 == 0 and x.data_ptr() % 16 == 0)


def _scan_desc(x, params, acts, hsum, w):
    N, K, D = x.shape
    d = _lib.AttnScanDesc()
    d.N, d.K, d.D = N, K, D
    d.x, d.x_sn, d.x_sk = x.data_ptr(), x.stride(0), x.stride(1)
    for q in range(2):
        d.w_ih[q], d.w_hh[q], d.b_ih[q], d.b_hh[q] = (t.data_ptr() for t in params[4 * q:4 * q + 4])
    d.acts, d.hsum, d.w = acts.data_ptr(), hsum.data_ptr(), w.data_ptr()
    return d


class AttentionScanFn(torch.autograd.Function):
    """kpgnn_attn_scan_fwd / _bwd: projection + recurrence + BPTT on the matrix cores, wave = (32 nodes, direction)."""

    @staticmethod
    def forward(ctx, x, *params):
        lib = _lib.load()
        N, K, D = x.shape
        refuse_dynamic_rows("the attention combine (its products run over every row)", N)
        dev = x.device
        params = [t.contiguous() for t in params]
        acts = torch.empty((((N + 31) // 32) * 2 * K * 20 * 64,), dtype=torch.float32, device=dev)
        hsum = torch.empty((2, N, K), dtype=torch.float32, device=dev)
        w = torch.empty((N, K), dtype=torch.float32, device=dev)
        out = torch.empty((N, D), dtype=torch.float32, device=dev)
        w_pad = torch.empty((64, D), dtype=torch.float32, device=dev)
        d = _scan_desc(x, params, acts, hsum, w)
        d.out, d.w_pad = out.data_ptr(), w_pad.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_attn_scan_fwd(ctypes.byref(d), _stream(x)), "kpgnn_attn_scan_fwd")
        ctx.save_for_backward(x, acts, w, w_pad, *params)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, acts, w, w_pad, *params = ctx.saved_tensors
        lib = _lib.load()
        gout = gout.contiguous()
        N, K, D = x.shape
        dev = x.device
        dx = torch.empty((N, K, D), dtype=torch.float32, device=dev)
        ds = torch.empty((N, K), dtype=torch.float32, device=dev)
        dgin = torch.empty((N * K, 64), dtype=torch.float32, device=dev)
        slab = torch.empty((((N + 31) // 32) * 512,), dtype=torch.float32, device=dev)
        dwhh_pad = torch.empty((2, 32, 8), dtype=torch.float32, device=dev)
        hsum = torch.empty((1,), dtype=torch.float32, device=dev)       # unused in backward
        d = _scan_desc(x, params, acts, hsum, w)
        d.w_pad = w_pad.data_ptr()
        d.gout, d.dx, d.ds, d.dgin = gout.data_ptr(), dx.data_ptr(), ds.data_ptr(), dgin.data_ptr()
        d.whh_slab, d.dwhh_pad = slab.data_ptr(), dwhh_pad.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_attn_scan_bwd(ctypes.byref(d), _stream(x)), "kpgnn_attn_scan_bwd")   # ds, dgin, dwhh_pad, dx
        xf = x.reshape(N * K, D)
        dw_pad, db_pad = _wgrad(dgin, xf, True)                          # [64, D], [64]
        dw = torch.empty((2, 4 * K, D), dtype=torch.float32, device=dev)
        db = torch.empty((2, 4 * K), dtype=torch.float32, device=dev)
        dwhh = torch.empty((2, 4 * K, K), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_attn_scan_unpad(dw_pad.data_ptr(), db_pad.data_ptr(), dwhh_pad.data_ptr(), dw.data_ptr(),
                                                 db.data_ptr(), dwhh.data_ptr(), K, D, _stream(x)), "kpgnn_attn_scan_unpad")
        return dx, dw[0], dwhh[0], db[0], db[0], dw[1], dwhh[1], db[1], db[1]


def attention_combine(x, lstm):
    """x [N,K,D] -> [N,D] with the parameters of the reference's nn.LSTM(D, K, bidirectional=True)."""
    N, K, D = x.shape
    if x.is_cuda and lstm.hidden_size == K:
        xs = _last_contig(x.float())
        if scan_applies(xs, K, D):
            return AttentionScanFn.apply(xs, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0,
                                         lstm.weight_ih_l0_reverse, lstm.weight_hh_l0_reverse,
                                         lstm.bias_ih_l0_reverse, lstm.bias_hh_l0_reverse)
    if x.is_cuda and K <= 16 and lstm.hidden_size == K and (D % 4 == 0 and D <= 256 or D <= 64):
        return AttentionCombineFn.apply(x, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0,
                                        lstm.weight_ih_l0_reverse, lstm.weight_hh_l0_reverse,
                                        lstm.bias_ih_l0_reverse, lstm.bias_hh_l0_reverse)
    if not x.is_cuda:
        raise _lib.KpgnnError("attention_combine needs CUDA/HIP tensors (no CPU fallback)")
    lstm.flatten_parameters()            # shapes outside the kernel's range: framework LSTM on the GPU
    score, _ = lstm(x)
    score = torch.softmax(score.sum(-1), dim=1).unsqueeze(-1)
    return (x * score).sum(1)
