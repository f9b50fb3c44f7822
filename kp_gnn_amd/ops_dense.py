"""Dense tail of the layers: training-mode BatchNorm1d (+ReLU, +residual) on the HIP kernels of bn.hip.

The nn.BatchNorm1d modules stay where the reference has them (state_dict keys `mlp.1.*`, `norms.l.module.*`);
only their training-mode arithmetic is routed here.  Eval mode (running statistics) uses torch's own GPU op."""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib
from .ops import _ptr, _stream


_bn_ws = {}


def _bn_workspace(lib, dev, C):
    """Persistent per-(device, C) scratch for the partial sums (no allocator round trip per call).  BatchNorm launches
    of one device are stream-ordered (one compute stream, or one captured graph), so they can share it."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), C)
    ent = _bn_ws.get(key)
    if ent is None:
        nb = int(lib.kpgnn_bn_workspace_bytes(C))
        ent = _bn_ws[key] = (torch.empty(nb, dtype=torch.uint8, device=dev), nb)
    return ent


class BatchNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, eps, momentum, relu, nbt=None):
        lib = _lib.load()
        x = x if x.stride(-1) == 1 else x.contiguous()
        N, C = x.shape
        dev = x.device
        z = torch.empty((N, C), dtype=torch.float32, device=dev)
        stats = torch.empty((2, C), dtype=torch.float32, device=dev)
        ws, ws_bytes = _bn_workspace(lib, dev, C)
        d = _lib.BnDesc()
        d.N, d.C, d.relu, d.eps, d.momentum = N, C, 1 if relu else 0, eps, momentum
        d.x, d.x_stride = x.data_ptr(), x.stride(0)
        d.gamma, d.beta = gamma.data_ptr(), beta.data_ptr()
        d.running_mean, d.running_var = _ptr(running_mean), _ptr(running_var)
        d.mean, d.invstd = stats[0].data_ptr(), stats[1].data_ptr()
        d.z, d.z_stride = z.data_ptr(), z.stride(0)
        if residual is not None:
            residual = residual if residual.stride(-1) == 1 else residual.contiguous()
            d.residual, d.r_stride = residual.data_ptr(), residual.stride(0)
        d.workspace, d.workspace_bytes = ws.data_ptr(), int(ws_bytes)
        d.num_batches_tracked = _ptr(nbt)
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_bn_fwd(ctypes.byref(d), _stream(x)), "kpgnn_bn_fwd")
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.relu = relu
        ctx.has_res = residual is not None
        return z

    @staticmethod
    def backward(ctx, dz):
        x, gamma, beta, stats = ctx.saved_tensors
        lib = _lib.load()
        dz = dz if dz.stride(-1) == 1 else dz.contiguous()
        N, C = x.shape
        dev = x.device
        dx = torch.empty((N, C), dtype=torch.float32, device=dev)
        dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
        ws, ws_bytes = _bn_workspace(lib, dev, C)
        d = _lib.BnBwdDesc()
        d.N, d.C, d.relu = N, C, 1 if ctx.relu else 0
        d.x, d.x_stride, d.dz, d.dz_stride = x.data_ptr(), x.stride(0), dz.data_ptr(), dz.stride(0)
        d.gamma, d.beta, d.mean, d.invstd = gamma.data_ptr(), beta.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr()
        d.dx, d.dx_stride = dx.data_ptr(), dx.stride(0)
        d.dgamma, d.dbeta = dgb[0].data_ptr(), dgb[1].data_ptr()
        d.workspace, d.workspace_bytes = ws.data_ptr(), int(ws_bytes)
        with torch.cuda.device(dev):
            _lib.check(lib.kpgnn_bn_bwd(ctypes.byref(d), _stream(x)), "kpgnn_bn_bwd")
        return dx, dgb[0], dgb[1], (dz if ctx.has_res else None), None, None, None, None, None, None


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
    if (O % 4 != 0 or I not in (32, 64, 104, 128) or O > 4096 or N < 1024 or not x.is_contiguous()
            or not w.is_contiguous() or x.data_ptr() % 16 or (bias is not None and bias.data_ptr() % 16)):
        return None
    y = torch.empty((N, O), dtype=torch.float32, device=x.device)
    d = _lib.LinearDesc()
    d.N, d.O, d.I = N, O, I
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
    return lin(x)


def batch_norm_act(x, bn, relu=False, residual=None):
    """nn.BatchNorm1d `bn` applied to x [N,C] (+ReLU) (+residual).  Training mode with batch statistics runs
    on the HIP kernels; everything else (eval, no affine, cumulative momentum, C > 256) on torch's GPU op."""
    use_hip = (bn.training and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and bn.affine
               and bn.momentum is not None and x.shape[1] <= 256 and x.shape[0] >= 1)
    if use_hip:
        rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
        nbt = bn.num_batches_tracked if bn.track_running_stats else None   # incremented inside the stats kernel
        return BatchNormAct.apply(x, bn.weight, bn.bias, residual, rm, rv, float(bn.eps), float(bn.momentum), relu, nbt)
    out = bn(x)
    if relu:
        out = F.relu(out)
    if residual is not None:
        out = out + residual
    return out


def mlp_linear_bn_relu_x2(mlp, h):
    """nn.Sequential(Linear, BatchNorm1d, ReLU, Linear, BatchNorm1d, ReLU) (KPGINplus.py:25-30, gine.py:31-38):
    the two GEMMs go to hipBLASLt, each BatchNorm+ReLU pair is one fused stats/apply on the HIP kernels."""
    h = batch_norm_act(linear(h, mlp[0]), mlp[1], relu=True)
    return batch_norm_act(linear(h, mlp[3]), mlp[4], relu=True)


# ------------------------------------------------------------------- KP-GIN per-hop MLP (+ geometric combine + projection)
class HopMlp(torch.autograd.Function):
    """relu(relu(s W1 + b1) W2 + b2) per hop (+ sum_k theta_k * . (+ combine_proj)), one HIP launch per direction
    (hop_mlp.hip)."""

    @staticmethod
    def forward(ctx, s, w1, b1, w2, b2, theta, wc, bc):
        lib = _lib.load()
        s = s.contiguous()
        N, K, DI = s.shape
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
