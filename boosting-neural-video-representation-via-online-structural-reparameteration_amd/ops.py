"""torch.autograd glue over the C ABI of liborn.so (include/orn.h).

Every function here takes CUDA (HIP) fp32 tensors, launches hand-written gfx950 kernels on the
current torch stream and returns torch tensors.  Nothing falls back to PyTorch math: a CPU tensor
or a missing library raises OrnError.
"""
import math
from ctypes import c_double, c_float, c_int, c_size_t

import torch

from . import _lib
from ._lib import check, lib, ptr, stream


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.OrnError('liborn ops need tensors on the GPU (there is no CPU path)')
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ---- A1 ------------------------------------------------------------------------------------
def pe_forward(pos: torch.Tensor, lbase: float, levels: int) -> torch.Tensor:
    """utils.py:121-129 on the device.  pos [B] -> [B, 2*levels]."""
    pos = _f32c(pos)
    pw = torch.tensor([lbase ** i for i in range(levels)], dtype=torch.float64).to(torch.float32).to(pos.device)
    out = torch.empty(pos.shape[0], 2 * levels, dtype=torch.float32, device=pos.device)
    check(lib().orn_pe_fwd(ptr(pos), c_int(pos.shape[0]), ptr(pw), c_int(levels), ptr(out), stream()), 'orn_pe_fwd')
    return out


# ---- A2 ------------------------------------------------------------------------------------
class StemFn(torch.autograd.Function):
    """SiLU(W1 SiLU(W0 e + b0) + b1) (model.py:186-188)."""

    @staticmethod
    def forward(ctx, embed, w0, b0, w1, b1):
        embed, w0, b0, w1, b1 = map(_f32c, (embed, w0, b0, w1, b1))
        B, E = embed.shape
        Hd, Nout = w0.shape[0], w1.shape[0]
        dev = embed.device
        pre1 = torch.empty(B, Hd, device=dev)
        h1 = torch.empty(B, Hd, device=dev)
        pre2 = torch.empty(B, Nout, device=dev)
        h2 = torch.empty(B, Nout, device=dev)
        check(lib().orn_stem_fwd(ptr(embed), ptr(w0), ptr(b0), ptr(w1), ptr(b1), B, E, Hd, Nout,
                                 ptr(pre1), ptr(h1), ptr(pre2), ptr(h2), stream()), 'orn_stem_fwd')
        ctx.save_for_backward(embed, w1, pre1, h1, pre2)
        ctx.dims = (B, E, Hd, Nout)
        return h2

    @staticmethod
    def backward(ctx, dh2):
        embed, w1, pre1, h1, pre2 = ctx.saved_tensors
        B, E, Hd, Nout = ctx.dims
        dev = embed.device
        dh2 = _f32c(dh2)
        dw0 = torch.empty(Hd, E, device=dev)
        db0 = torch.empty(Hd, device=dev)
        dw1 = torch.empty(Nout, Hd, device=dev)
        db1 = torch.empty(Nout, device=dev)
        ws = torch.empty(B * Nout + 2 * B * Hd + 256 * B * Hd, device=dev)
        check(lib().orn_stem_bwd(ptr(embed), ptr(w1), ptr(pre1), ptr(h1), ptr(pre2), ptr(dh2), B, E, Hd, Nout,
                                 ptr(dw0), ptr(db0), ptr(dw1), ptr(db1), ptr(ws), stream()), 'orn_stem_bwd')
        return None, dw0, db0, dw1, db1


# ---- A3 ------------------------------------------------------------------------------------
class ErbMergeFn(torch.autograd.Function):
    """get_equivalent_kernel_bias (model.py:450-516): 9 branch tensors -> (Wf, bf)."""

    @staticmethod
    def forward(ctx, w3x3, b3x3, w3x1, b3x1, w1x3, b1x3, w1, w2, w3):
        ts = list(map(_f32c, (w3x3, b3x3, w3x1, b3x1, w1x3, b1x3, w1, w2, w3)))
        O, C = ts[0].shape[0], ts[0].shape[1]
        dev = ts[0].device
        T = torch.empty(O, C, 3, 3, device=dev)
        wf = torch.empty(O, C, 3, 3, device=dev)
        bf = torch.empty(O, device=dev)
        check(lib().orn_erb_merge_fwd(*[ptr(t) for t in ts], C, O, ptr(T), ptr(wf), ptr(bf), stream()),
              'orn_erb_merge_fwd')
        ctx.save_for_backward(ts[6], ts[7], ts[8], T)
        ctx.dims = (C, O)
        return wf, bf

    @staticmethod
    def backward(ctx, g, dbf):
        w1, w2, w3, T = ctx.saved_tensors
        C, O = ctx.dims
        dev = w1.device
        g = _f32c(g) if g is not None else torch.zeros(O, C, 3, 3, device=dev)
        dbf = _f32c(dbf) if dbf is not None else torch.zeros(O, device=dev)
        d3x3 = torch.empty(O, C, 3, 3, device=dev)
        db3x3 = torch.empty(O, device=dev)
        d3x1 = torch.empty(O, C, 3, 1, device=dev)
        db3x1 = torch.empty(O, device=dev)
        d1x3 = torch.empty(O, C, 1, 3, device=dev)
        db1x3 = torch.empty(O, device=dev)
        dw1 = torch.empty(2 * C, C, 1, 1, device=dev)
        dw2 = torch.empty(O, 2 * C, 3, 3, device=dev)
        dw3 = torch.empty(O, O, 1, 1, device=dev)
        nb = lib().orn_erb_merge_bwd_ws_bytes(C, O)
        ws = _ws(nb, dev)
        check(lib().orn_erb_merge_bwd(ptr(g), ptr(dbf), ptr(w1), ptr(w2), ptr(w3), ptr(T), C, O, ptr(d3x3), ptr(db3x3),
                                      ptr(d3x1), ptr(db3x1), ptr(d1x3), ptr(db1x3), ptr(dw1), ptr(dw2), ptr(dw3),
                                      ptr(ws), c_size_t(ws.numel()), stream()), 'orn_erb_merge_bwd')
        return d3x3, db3x3, d3x1, db3x1, d1x3, db1x3, dw1, dw2, dw3


# ---- A4 ------------------------------------------------------------------------------------
class ConvPsSiluFn(torch.autograd.Function):
    """conv3x3(pad 1) + bias -> PixelShuffle(s) -> SiLU (model.py:539,567)."""

    @staticmethod
    def forward(ctx, x, wf, bf, s):
        x, wf, bf = _f32c(x), _f32c(wf), _f32c(bf)
        B, C, H, W = x.shape
        O = wf.shape[0]
        if wf.shape[1] != C or tuple(wf.shape[2:]) != (3, 3) or O % (s * s):
            raise _lib.OrnError(f'conv3x3_ps_silu: bad shapes x={tuple(x.shape)} w={tuple(wf.shape)} s={s}')
        dev = x.device
        need_grad = any(ctx.needs_input_grad[:3])
        a = torch.empty(B, O // (s * s), H * s, W * s, device=dev)
        z = torch.empty_like(a) if need_grad else None
        check(lib().orn_conv3x3_ps_silu_fwd(ptr(x), ptr(wf), ptr(bf), B, C, O, H, W, s, ptr(z), ptr(a), stream()),
              'orn_conv3x3_ps_silu_fwd')
        if need_grad:
            ctx.save_for_backward(x, wf, z)
        ctx.dims = (B, C, O, H, W, s)
        return a

    @staticmethod
    def backward(ctx, da):
        x, wf, z = ctx.saved_tensors
        B, C, O, H, W, s = ctx.dims
        dev = x.device
        da = _f32c(da)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dwf = torch.empty_like(wf)
        dbf = torch.empty(O, device=dev)
        nb = lib().orn_conv3x3_ps_silu_bwd_ws_bytes(B, C, O, H, W)
        ws = _ws(nb, dev)
        check(lib().orn_conv3x3_ps_silu_bwd(ptr(x), ptr(wf), ptr(z), ptr(da), B, C, O, H, W, s, ptr(dx), ptr(dwf),
                                            ptr(dbf), ptr(ws), c_size_t(ws.numel()), stream()),
              'orn_conv3x3_ps_silu_bwd')
        return dx, dwf, dbf, None


# ---- A5 ------------------------------------------------------------------------------------
class HeadFn(torch.autograd.Function):
    """1x1 conv -> (tanh+1)/2 or sigmoid (model.py:621-622)."""

    @staticmethod
    def forward(ctx, a, w, b, sigmoid):
        a, w, b = _f32c(a), _f32c(w), _f32c(b)
        B, C, H, W = a.shape
        out = torch.empty(B, 3, H, W, device=a.device)
        check(lib().orn_head_fwd(ptr(a), ptr(w), ptr(b), B, C, H, W, int(bool(sigmoid)), ptr(out), stream()), 'orn_head_fwd')
        ctx.save_for_backward(a, w, out)
        ctx.sigmoid = int(bool(sigmoid))
        return out

    @staticmethod
    def backward(ctx, dout):
        a, w, out = ctx.saved_tensors
        B, C, H, W = a.shape
        dev = a.device
        dout = _f32c(dout)
        da = torch.empty_like(a)
        dw = torch.empty_like(w)
        db = torch.empty(3, device=dev)
        nb = lib().orn_head_bwd_ws_bytes(B, C, H, W)
        ws = _ws(nb, dev)
        check(lib().orn_head_bwd(ptr(a), ptr(w), ptr(out), ptr(dout), B, C, H, W, ctx.sigmoid, ptr(da), ptr(dw), ptr(db),
                                 ptr(ws), c_size_t(ws.numel()), stream()), 'orn_head_bwd')
        return da, dw, db, None


# ---- A7 / A10 ------------------------------------------------------------------------------
def loss_stats(pred, target, loss_type: str = 'Fusion6', want_grad: bool = True, loss_scale: float = 1.0):
    """-> (stats[8] device tensor, dpred or None).  stats = [loss, L1, MSE, SSIM, PSNR, 0, 0, 0]."""
    if loss_type not in _lib.LOSS_TYPES:
        raise NotImplementedError(f'loss_type {loss_type!r}: only L2, L1 and Fusion6 are built (utils.py:139-189)')
    pred, target = _f32c(pred), _f32c(target)
    if pred.shape != target.shape or pred.dim() != 4:
        raise _lib.OrnError(f'loss: shapes {tuple(pred.shape)} vs {tuple(target.shape)}')
    B, Ch, H, W = pred.shape
    dev = pred.device
    stats = torch.empty(8, device=dev)
    dpred = torch.empty_like(pred) if want_grad else None
    nb = lib().orn_loss_ws_bytes(B, Ch, H, W)
    ws = _ws(nb, dev)
    check(lib().orn_loss_fwd_bwd(ptr(pred), ptr(target), B, Ch, H, W, _lib.LOSS_TYPES[loss_type], c_float(loss_scale),
                                 ptr(stats), ptr(dpred), ptr(ws), c_size_t(ws.numel()), stream()), 'orn_loss_fwd_bwd')
    return stats, dpred


class LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, loss_type):
        stats, dpred = loss_stats(pred, target.detach(), loss_type, want_grad=True)
        ctx.save_for_backward(dpred)
        return stats[0].clone()

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None


# ---- A9 ------------------------------------------------------------------------------------
def adam_step_(p, g, m, v, lr: float, step: int, beta1: float = 0.5, beta2: float = 0.999, eps: float = 1e-8):
    """In-place Adam over flat fp32 arenas (main_train.py:196,250)."""
    for t in (p, g, m, v):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise _lib.OrnError('adam_step_: arenas must be contiguous CUDA fp32 tensors')
    n = p.numel()
    check(lib().orn_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), c_size_t(n), c_double(lr), c_double(beta1), c_double(beta2),
                              c_double(eps), c_int(step), stream()), 'orn_adam_step')


# ---- N3 ------------------------------------------------------------------------------------
def ms_ssim(pred, target) -> torch.Tensor:
    """pytorch_msssim.ms_ssim(pred, target, data_range=1, size_average=True) as utils.py:205 calls it."""
    pred, target = _f32c(pred.detach()), _f32c(target.detach())
    B, Ch, H, W = pred.shape
    out = torch.empty(1, device=pred.device)
    nb = lib().orn_msssim_ws_bytes(B, Ch, H, W)
    ws = _ws(nb, pred.device)
    check(lib().orn_msssim(ptr(pred), ptr(target), B, Ch, H, W, ptr(out), ptr(ws), c_size_t(ws.numel()), stream()), 'orn_msssim')
    return out[0]
