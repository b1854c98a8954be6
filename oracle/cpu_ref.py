"""CPU oracle for the Online-RepNeRV training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the
checker / the timed CPU baseline.  The shipped path (the HIP library behind ``include/orn.h``)
never routes through this module.

It is a functional (stateless) restatement of the reference's per-frame training step on torch-CPU
ops, each function citing the reference lines it follows (paths relative to ``/root/reference``).

Parity status
-------------
* Everything except SSIM is pinned by golden vectors captured from the reference's own
  ``model.py`` / ``utils.py`` in the build container (``tools/make_golden.py`` ->
  ``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).
* ``ssim`` / ``ms_ssim``: the reference calls the third-party ``pytorch_msssim==0.2.1``
  (``requirements.txt:3``, ``utils.py:9``), which is absent from the reference tree and from this
  image.  They are restated here from the package's published algorithm (Wang et al. 2004;
  11-tap Gaussian, sigma 1.5, separable *valid* filtering, K=(0.01,0.03)) -> **parity unpinned**
  (definition-level) for SSIM / MS-SSIM and therefore for the SSIM term of Fusion6.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# A1  positional encoding                                              utils.py:110-129
# --------------------------------------------------------------------------------------------

def pe_params(pe_embed: str) -> Tuple[float, int]:
    """'1.25_40' -> (1.25, 40); embed_length = 2*levels (utils.py:117-119)."""
    lbase, levels = [float(x) for x in pe_embed.split('_')]
    return lbase, int(levels)


def positional_encoding(pos: torch.Tensor, lbase: float, levels: int) -> torch.Tensor:
    """utils.py:121-129.  ``pos`` fp32 [B] -> fp32 [B, 2*levels].

    The argument is formed in fp32, left to right, with Python-double scalars that torch rounds to
    fp32 before multiplying: fp32(fp32(pos * fp32(lbase**i)) * fp32(pi)).  That rounding is
    load-bearing: arguments reach 1.9e4 rad where one fp32 ulp is 2e-3 rad.
    """
    pe_list = []
    for i in range(levels):
        temp_value = pos * lbase ** (i) * math.pi
        pe_list += [torch.sin(temp_value), torch.cos(temp_value)]
    return torch.stack(pe_list, 1)


def pe_arguments(pos: np.ndarray, lbase: float, levels: int) -> np.ndarray:
    """The fp32 sin/cos arguments of utils.py:127 as an explicit numpy restatement [B, levels]."""
    pos = np.asarray(pos, dtype=np.float32)
    out = np.empty((pos.shape[0], levels), dtype=np.float32)
    for i in range(levels):
        out[:, i] = (pos * np.float32(lbase ** i)).astype(np.float32) * np.float32(math.pi)
    return out


# --------------------------------------------------------------------------------------------
# A2  stem MLP                                                          model.py:174-188, 612-613
# --------------------------------------------------------------------------------------------

def stem_forward(embed: torch.Tensor, w0, b0, w1, b1) -> torch.Tensor:
    """Linear+SiLU after *every* Linear, including the last (model.py:186-188)."""
    h = F.silu(F.linear(embed, w0, b0))
    return F.silu(F.linear(h, w1, b1))


# --------------------------------------------------------------------------------------------
# A3  ERB online re-parameterisation                                    model.py:450-516
# --------------------------------------------------------------------------------------------

ERB_KEYS = (
    'rbr_3x3_branch.weight', 'rbr_3x3_branch.bias',
    'rbr_3x1_branch.weight', 'rbr_3x1_branch.bias',
    'rbr_1x3_branch.weight', 'rbr_1x3_branch.bias',
    'rbr_1x1_3x3_1x1_branch_1x1_1.weight',
    'rbr_1x1_3x3_1x1_branch_3x3.weight',
    'rbr_1x1_3x3_1x1_branch_1x1_2.weight',
)


def erb_merge(w3x3, b3x3, w3x1, b3x1, w1x3, b1x3, w1, w2, w3):
    """get_equivalent_kernel_bias (model.py:450-478) with the same ATen ops as the reference.

    w3x3 [O,C,3,3]; w3x1 [O,C,3,1]; w1x3 [O,C,1,3]; w1 [2C,C,1,1]; w2 [O,2C,3,3]; w3 [O,O,1,1].
    """
    # _fuse_1x3_3x1_branch, model.py:495-496
    k13_31 = F.pad(w1x3, (0, 0, 1, 1)) + F.pad(w3x1, (1, 1, 0, 0))
    b13_31 = b1x3 + b3x1
    # _fuse_1x1_3x3_1x1_branch, model.py:510-515
    tmp = F.conv2d(w2, w1.permute(1, 0, 2, 3))
    k0 = tmp.permute(2, 3, 0, 1)
    k1 = w3.permute(2, 3, 0, 1).repeat(3, 3, 1, 1)
    kseq = torch.matmul(k1, k0).permute(2, 3, 0, 1)
    # model.py:475-476 (association order matters at 1 ulp)
    fused_kernel = w3x3 + k13_31 + kseq
    fused_bias = b3x3 + b13_31
    return fused_kernel, fused_bias


def erb_merge_ordered_np(w3x3, b3x3, w3x1, b3x1, w1x3, b1x3, w1, w2, w3):
    """The *specified* summation order the HIP merge kernel implements, in pure numpy fp32.

    T[m,c,i,j] = fma-chain over k=0..2C-1 of W2[m,k,i,j]*W1[k,c]       (model.py:510)
    S[o,c,i,j] = fma-chain over m=0..O-1  of W3[o,m]*T[m,c,i,j]        (model.py:513-515)
    Wf = (W3x3 + (P(w1x3) + P(w3x1))) + S ; bf = b3x3 + (b1x3 + b3x1)  (model.py:475-476,495-496)

    numpy has no fused multiply-add, so the chain is evaluated in float64 products rounded once per
    step: fma(a,b,c) == fp32(fp64(a)*fp64(b) + fp64(c)) exactly whenever the fp64 sum is exact or
    rounds identically -- a*b is exact in fp64 (24+24 bits), and the fp64 add of a 48-bit product
    and a 24-bit addend can need >53 bits only when exponents differ by >5 bits, where double
    rounding could differ in rare ties.  ``oracle/merge_ref.c`` (true ``fmaf``) is the authority;
    this function exists so small cases run without the C build and is checked against it.
    Returns (Wf, bf, T).
    """
    w1 = np.asarray(w1, np.float32).reshape(w1.shape[0], w1.shape[1])       # [2C, C]
    w2 = np.asarray(w2, np.float32)                                         # [O, 2C, 3, 3]
    w3 = np.asarray(w3, np.float32).reshape(w3.shape[0], w3.shape[1])       # [O, O]
    O, K2 = w2.shape[0], w2.shape[1]
    C = w1.shape[1]
    T = np.zeros((O, C, 3, 3), np.float32)
    for k in range(K2):
        prod = w2[:, k, None, :, :].astype(np.float64) * w1[k, None, :, None, None].astype(np.float64)
        T = (prod + T.astype(np.float64)).astype(np.float32)
    S = np.zeros((O, C, 3, 3), np.float32)
    for m in range(O):
        prod = w3[:, m, None, None, None].astype(np.float64) * T[m][None].astype(np.float64)
        S = (prod + S.astype(np.float64)).astype(np.float32)
    p13 = np.zeros((O, C, 3, 3), np.float32)
    p13[:, :, 1, :] = np.asarray(w1x3, np.float32)[:, :, 0, :]
    p31 = np.zeros((O, C, 3, 3), np.float32)
    p31[:, :, :, 1] = np.asarray(w3x1, np.float32)[:, :, :, 0]
    wf = (np.asarray(w3x3, np.float32) + (p13 + p31)) + S
    bf = np.asarray(b3x3, np.float32) + (np.asarray(b1x3, np.float32) + np.asarray(b3x1, np.float32))
    return wf, bf, T


def erb_merge_backward_closed_form(g, dbf, w1, w2, w3, T=None):
    """Closed-form backward of the merge, torch tensors, any float dtype.

    g = dL/dWf [O,C,3,3], dbf = dL/dbf [O].  Returns a dict keyed like ERB_KEYS.
    Verified against autograd of ``erb_merge`` in tests/test_oracle_golden.py.
    """
    O, C = g.shape[0], g.shape[1]
    w1m = w1.reshape(2 * C, C)
    w3m = w3.reshape(O, O)
    if T is None:
        T = torch.einsum('mkij,kc->mcij', w2, w1m)
    dW3 = torch.einsum('ocij,mcij->om', g, T)
    dT = torch.einsum('om,ocij->mcij', w3m, g)
    dW2 = torch.einsum('mcij,kc->mkij', dT, w1m)
    dW1 = torch.einsum('mkij,mcij->kc', w2, dT)
    return {
        'rbr_3x3_branch.weight': g.clone(),
        'rbr_3x3_branch.bias': dbf.clone(),
        'rbr_3x1_branch.weight': g[:, :, :, 1:2].clone(),
        'rbr_3x1_branch.bias': dbf.clone(),
        'rbr_1x3_branch.weight': g[:, :, 1:2, :].clone(),
        'rbr_1x3_branch.bias': dbf.clone(),
        'rbr_1x1_3x3_1x1_branch_1x1_1.weight': dW1.reshape(2 * C, C, 1, 1),
        'rbr_1x1_3x3_1x1_branch_3x3.weight': dW2,
        'rbr_1x1_3x3_1x1_branch_1x1_2.weight': dW3.reshape(O, O, 1, 1),
    }


# --------------------------------------------------------------------------------------------
# A4  NeRVBlock forward                                                 model.py:518-567
# --------------------------------------------------------------------------------------------

def block_forward(x, wf, bf, stride: int):
    """conv3x3(pad 1) -> PixelShuffle(stride) -> SiLU (norm = Identity) (model.py:539,567)."""
    y = F.conv2d(x, wf, bf, stride=1, padding=1)
    return F.silu(F.pixel_shuffle(y, stride))


# --------------------------------------------------------------------------------------------
# A5  head                                                              model.py:617-625
# --------------------------------------------------------------------------------------------

def head_forward(a, w, b, sigmoid: bool = False):
    u = F.conv2d(a, w, b)
    return torch.sigmoid(u) if sigmoid else (torch.tanh(u) + 1) * 0.5


# --------------------------------------------------------------------------------------------
# Whole Generator (ERB / vanilla / deploy), single_res                  model.py:571-625
# --------------------------------------------------------------------------------------------

def layer_geometry(fc_hw_dim: str, strides: Sequence[int], expansion: float, reduction: int,
                   lower_width: int) -> List[Dict[str, int]]:
    """Per-layer (C, new_ngf, O, s, H_in, W_in) following model.py:583-595 (num_blocks = 1)."""
    fc_h, fc_w, fc_dim = [int(x) for x in fc_hw_dim.split('_')]
    ngf, h, w = fc_dim, fc_h, fc_w
    out = []
    for i, s in enumerate(strides):
        if i == 0:
            new_ngf = int(ngf * expansion)
        else:
            new_ngf = max(ngf // (1 if s == 1 else reduction), lower_width)
        out.append(dict(C=ngf, new_ngf=new_ngf, O=new_ngf * s * s, s=s, H=h, W=w))
        ngf, h, w = new_ngf, h * s, w * s
    return out


def generator_forward(sd: Dict[str, torch.Tensor], embed: torch.Tensor, fc_hw_dim: str,
                      strides: Sequence[int], branch_type: str = 'ERB', sigmoid: bool = False,
                      return_intermediates: bool = False):
    """Generator.forward (model.py:611-625) for --single_res, from a reference-layout state dict.

    ``branch_type``: 'ERB' (train-time branches, online merge every call), 'NeRV_vanilla'
    (``layers.N.branch``) or 'deploy' (``layers.N.rbr_reparam``).
    """
    fc_h, fc_w, fc_dim = [int(x) for x in fc_hw_dim.split('_')]
    out = stem_forward(embed, sd['stem.0.weight'], sd['stem.0.bias'], sd['stem.2.weight'], sd['stem.2.bias'])
    out = out.view(out.size(0), fc_dim, fc_h, fc_w)
    inter = [out]
    n = len(strides)
    for i, s in enumerate(strides):
        p = f'layers.{i}.'
        if branch_type == 'ERB':
            wf, bf = erb_merge(*[sd[p + k] for k in ERB_KEYS])
        elif branch_type == 'NeRV_vanilla':
            wf, bf = sd[p + 'branch.weight'], sd[p + 'branch.bias']
        else:
            wf, bf = sd[p + 'rbr_reparam.weight'], sd[p + 'rbr_reparam.bias']
        out = block_forward(out, wf, bf, s)
        inter.append(out)
    img = head_forward(out, sd[f'head_layers.{n - 1}.weight'], sd[f'head_layers.{n - 1}.bias'], sigmoid)
    if return_intermediates:
        return [img], inter
    return [img]


# --------------------------------------------------------------------------------------------
# A7  loss (Fusion6 and the trivial ones)                               utils.py:139-189
# --------------------------------------------------------------------------------------------

def _gauss_1d(size: int = 11, sigma: float = 1.5, dtype=torch.float32) -> torch.Tensor:
    """pytorch_msssim 0.2.1 ``_fspecial_gauss_1d`` (published algorithm; parity unpinned)."""
    coords = torch.arange(size, dtype=dtype) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _gaussian_filter(x: torch.Tensor, win: torch.Tensor) -> torch.Tensor:
    """Separable *valid* depthwise filtering along H then W."""
    c = x.shape[1]
    k = win.numel()
    out = F.conv2d(x, win.view(1, 1, k, 1).repeat(c, 1, 1, 1), groups=c)
    out = F.conv2d(out, win.view(1, 1, 1, k).repeat(c, 1, 1, 1), groups=c)
    return out


def ssim_maps(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0,
              K: Tuple[float, float] = (0.01, 0.03)):
    """ssim_map, cs_map of pytorch_msssim ``_ssim`` (published algorithm; parity unpinned)."""
    c1 = (K[0] * data_range) ** 2
    c2 = (K[1] * data_range) ** 2
    win = _gauss_1d(dtype=x.dtype)
    mu1 = _gaussian_filter(x, win)
    mu2 = _gaussian_filter(y, win)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    sigma1_sq = _gaussian_filter(x * x, win) - mu1_sq
    sigma2_sq = _gaussian_filter(y * y, win) - mu2_sq
    sigma12 = _gaussian_filter(x * y, win) - mu1_mu2
    cs_map = (2 * sigma12 + c2) / (sigma1_sq + sigma2_sq + c2)
    ssim_map = ((2 * mu1_mu2 + c1) / (mu1_sq + mu2_sq + c1)) * cs_map
    return ssim_map, cs_map


def ssim(x, y, data_range: float = 1.0, size_average: bool = True):
    """``pytorch_msssim.ssim(X, Y, data_range=1, size_average=True)`` as called at utils.py:160."""
    ssim_map, _ = ssim_maps(x, y, data_range)
    per_channel = torch.flatten(ssim_map, 2).mean(-1)
    return per_channel.mean() if size_average else per_channel.mean(1)


_MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def ms_ssim(x, y, data_range: float = 1.0, size_average: bool = True):
    """``pytorch_msssim.ms_ssim`` (5 scales, avg_pool2d(2) with padding = size%2; parity unpinned)."""
    weights = torch.tensor(_MS_WEIGHTS, dtype=x.dtype)
    mcs = []
    levels = weights.numel()
    for i in range(levels):
        ssim_map, cs_map = ssim_maps(x, y, data_range)
        ssim_pc = torch.flatten(ssim_map, 2).mean(-1)
        cs = torch.flatten(cs_map, 2).mean(-1)
        if i < levels - 1:
            mcs.append(torch.relu(cs))
            padding = [s % 2 for s in x.shape[2:]]
            x = F.avg_pool2d(x, kernel_size=2, padding=padding)
            y = F.avg_pool2d(y, kernel_size=2, padding=padding)
    ssim_pc = torch.relu(ssim_pc)
    mcs_and_ssim = torch.stack(mcs + [ssim_pc], dim=0)            # [levels, B, C]
    val = torch.prod(mcs_and_ssim ** weights.view(-1, 1, 1), dim=0)
    return val.mean() if size_average else val.mean(1)


def loss_fn(pred, target, loss_type: str = 'Fusion6'):
    """utils.py:139-189, the variants BASELINE configs use (Fusion6) plus the SSIM-free ones."""
    target = target.detach()
    if loss_type == 'L2':
        return F.mse_loss(pred, target, reduction='none').mean()
    if loss_type == 'L1':
        return torch.mean(torch.abs(pred - target))
    if loss_type == 'SSIM':
        return 1 - ssim(pred, target, data_range=1, size_average=True)
    if loss_type == 'Fusion6':
        return 0.7 * torch.mean(torch.abs(pred - target)) + 0.3 * (1 - ssim(pred, target, data_range=1, size_average=True))
    if loss_type == 'Fusion7':
        return 0.7 * F.mse_loss(pred, target) + 0.3 * torch.mean(torch.abs(pred - target))
    if loss_type == 'Fusion8':
        return 0.5 * F.mse_loss(pred, target) + 0.5 * torch.mean(torch.abs(pred - target))
    raise NotImplementedError(loss_type)


def fusion6_grad_closed_form(p: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """dL/dp of Fusion6 from SURVEY Appendix B (checked against autograd in the tests)."""
    n = p.numel()
    win = _gauss_1d(dtype=p.dtype)
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m, mu = _gaussian_filter(p, win), _gaussian_filter(t, win)
    q, r = _gaussian_filter(p * p, win), _gaussian_filter(p * t, win)
    sp = q - m * m
    st = _gaussian_filter(t * t, win) - mu * mu
    spt = r - m * mu
    a1, a2 = 2 * m * mu + c1, 2 * spt + c2
    b1, b2 = m * m + mu * mu + c1, sp + st + c2
    s = a1 * a2 / (b1 * b2)
    ds_dm = 2 * mu * (a2 - a1) / (b1 * b2) - 2 * m * s / b1 + 2 * m * s / b2
    ds_dq = -s / b2
    ds_dr = 2 * a1 / (b1 * b2)
    nmap = s.numel()

    def adj(gmap):   # adjoint of the valid separable filter: zero-pad 10, same (symmetric) taps
        c = gmap.shape[1]
        k = win.numel()
        g = F.conv2d(gmap, win.view(1, 1, k, 1).repeat(c, 1, 1, 1), groups=c, padding=(k - 1, 0))
        return F.conv2d(g, win.view(1, 1, 1, k).repeat(c, 1, 1, 1), groups=c, padding=(0, k - 1))

    dssim = (adj(ds_dm) + 2 * p * adj(ds_dq) + t * adj(ds_dr)) / nmap
    return 0.7 * torch.sign(p - t) / n - 0.3 * dssim


# --------------------------------------------------------------------------------------------
# A10 PSNR, A8 LR schedule, A9 Adam                         utils.py:191-199, 240-259; main_train.py:196
# --------------------------------------------------------------------------------------------

def psnr_fn(output_list, target_list):
    """utils.py:191-199: batch-mean MSE -> -10 log10, expanded to [B, stages]."""
    psnr_list = []
    for output, target in zip(output_list, target_list):
        l2 = F.mse_loss(output.detach(), target.detach(), reduction='mean')
        psnr = -10 * torch.log10(l2)
        psnr_list.append(psnr.view(1, 1).expand(output.size(0), -1))
    return torch.cat(psnr_list, dim=1)


def adjust_lr_value(cur_epoch: int, cur_iter: int, data_size: int, lr: float, epochs: int,
                    warmup: int, lr_type: str = 'cosine', lr_steps: Sequence[float] = ()) -> float:
    """utils.py:240-259 in Python doubles; ``warmup`` is already int(warmup_ratio*epochs)
    (main_train.py:111)."""
    e = cur_epoch + (float(cur_iter) / data_size)
    if lr_type == 'cosine':
        lr_mult = 0.5 * (math.cos(math.pi * (e - warmup) / (epochs - warmup)) + 1.0)
    elif lr_type == 'step':
        lr_mult = 0.1 ** (sum(e >= np.array(lr_steps)))
    elif lr_type in ('const', 'plateau'):
        lr_mult = 1
    else:
        raise NotImplementedError
    if e < warmup:
        lr_mult = 0.1 + 0.9 * e / warmup
    return lr * lr_mult


def adam_step(p, g, m, v, step: int, lr: float, beta1: float = 0.5, beta2: float = 0.999,
              eps: float = 1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad) as main_train.py:196 configures it.
    ``step`` is the 1-based global step count.  In place on p, m, v.  Same algebra as
    torch's single-tensor path: denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= lr/(1-b1^t) * m/denom."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


# --------------------------------------------------------------------------------------------
# A11 one training step (autograd over the functional forward)         main_train.py:229-254
# --------------------------------------------------------------------------------------------

def train_step(sd: Dict[str, torch.Tensor], adam_m, adam_v, step: int, lr: float, embed, target,
               fc_hw_dim: str, strides, branch_type: str = 'ERB', loss_type: str = 'Fusion6',
               beta1: float = 0.5):
    """One optimiser step on the state dict ``sd`` (leaf tensors, updated in place).

    Returns (loss, psnr, grads dict).  ``adam_m`` / ``adam_v`` are dicts keyed like ``sd``.
    """
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    out = generator_forward(params, embed, fc_hw_dim, strides, branch_type)[0]
    loss = loss_fn(out, target, loss_type)
    loss.backward()
    psnr = psnr_fn([out], [target])
    grads = {}
    with torch.no_grad():
        for k in sd:
            g = params[k].grad
            if g is None:
                continue
            grads[k] = g
            adam_step(sd[k], g, adam_m[k], adam_v[k], step, lr, beta1)
    return loss.detach(), psnr, grads


def init_state_dict(embed_length: int, stem_dim_num: str, fc_hw_dim: str, strides, expansion, reduction,
                    lower_width, branch_type: str = 'ERB', seed: int = 1) -> Dict[str, torch.Tensor]:
    """Random init with PyTorch's default Linear/Conv2d initialisers (kaiming_uniform a=sqrt(5) ->
    U(+-1/sqrt(fan_in)) for weight and bias), drawn in the reference's module construction order
    (model.py:575-608) under ``torch.manual_seed(seed)`` (main_train.py:162) so the result equals
    ``Generator(...).state_dict()`` of the reference for the same seed (checked by the golden test).
    """
    g = torch.Generator().manual_seed(seed)

    def uni(shape, fan_in):
        bound = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def kaiming_uniform(shape, fan_in):
        # init.kaiming_uniform_(w, a=sqrt(5)): bound = sqrt(6/((1+5)*fan_in)) = 1/sqrt(fan_in)
        return torch.empty(shape).uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in), generator=g)

    def linear(prefix, fin, fout, sd):
        sd[prefix + '.weight'] = kaiming_uniform((fout, fin), fin)
        sd[prefix + '.bias'] = kaiming_uniform((fout,), fin)

    def conv(prefix, cin, cout, kh, kw, sd, bias=True):
        fan_in = cin * kh * kw
        sd[prefix + '.weight'] = kaiming_uniform((cout, cin, kh, kw), fan_in)
        if bias:
            sd[prefix + '.bias'] = kaiming_uniform((cout,), fan_in)

    sd: Dict[str, torch.Tensor] = {}
    stem_dim, stem_num = [int(x) for x in stem_dim_num.split('_')]
    fc_h, fc_w, fc_dim = [int(x) for x in fc_hw_dim.split('_')]
    dims = [embed_length] + [stem_dim] * stem_num + [fc_h * fc_w * fc_dim]
    for i in range(len(dims) - 1):
        linear(f'stem.{2 * i}', dims[i], dims[i + 1], sd)
    geo = layer_geometry(fc_hw_dim, strides, expansion, reduction, lower_width)
    for i, L in enumerate(geo):
        C, O = L['C'], L['O']
        p = f'layers.{i}.'
        if branch_type == 'ERB':
            conv(p + 'rbr_3x3_branch', C, O, 3, 3, sd)
            conv(p + 'rbr_3x1_branch', C, O, 3, 1, sd)
            conv(p + 'rbr_1x3_branch', C, O, 1, 3, sd)
            conv(p + 'rbr_1x1_3x3_1x1_branch_1x1_1', C, 2 * C, 1, 1, sd, bias=False)
            conv(p + 'rbr_1x1_3x3_1x1_branch_3x3', 2 * C, O, 3, 3, sd, bias=False)
            conv(p + 'rbr_1x1_3x3_1x1_branch_1x1_2', O, O, 1, 1, sd, bias=False)
        elif branch_type == 'NeRV_vanilla':
            conv(p + 'branch', C, O, 3, 3, sd)
        else:
            conv(p + 'rbr_reparam', C, O, 3, 3, sd)
        if i == len(geo) - 1:       # --single_res: only the last stage has a head (model.py:599-604)
            conv(f'head_layers.{i}', L['new_ngf'], 3, 1, 1, sd)
    return sd


def synthetic_video(frames: int, h: int, w: int, seed: int = 1234) -> torch.Tensor:
    """Synthetic smooth-ish video [frames,3,h,w] fp32 in [0,1] (SURVEY 8d recipe):
    V[k] = clip(0.5 + 0.25*sin(2pi(fx*x + fy*y + k/frames)) summed over 4 (fx,fy) per channel / 4*...
    + 0.1*U(-1,1)).  Deterministic from ``seed``; used by bench.py and the engine tests."""
    g = torch.Generator().manual_seed(seed)
    ys = torch.linspace(0, 1, h).view(1, 1, h, 1)
    xs = torch.linspace(0, 1, w).view(1, 1, 1, w)
    ks = (torch.arange(frames, dtype=torch.float32) / frames).view(frames, 1, 1, 1)
    v = torch.full((frames, 3, h, w), 0.5)
    for _ in range(4):
        fx = (torch.rand(3, generator=g) * 6 + 0.5).view(1, 3, 1, 1)
        fy = (torch.rand(3, generator=g) * 6 + 0.5).view(1, 3, 1, 1)
        ph = torch.rand(3, generator=g).view(1, 3, 1, 1)
        v = v + (0.25 / 4) * torch.sin(2 * math.pi * (fx * xs + fy * ys + ks + ph))
    noise = torch.rand((frames, 3, h, w), generator=g) * 2 - 1
    return (v + 0.1 * noise).clamp_(0, 1).contiguous()
