"""GPU tests of the bf16 MFMA fast path (C-ABI hooks orn_conv3x3_ps_silu_{fwd,bwd}_bf16) against the
fp32 oracle.  Tolerances are bf16 tolerances: inputs/weights rounded to 8 significant bits, fp32
accumulation; stated per assertion."""
import math
from ctypes import c_size_t

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def orn():
    import orn_amd
    from orn_amd import ops, model, utils, engine  # noqa: F401
    orn_amd._lib.lib()
    return orn_amd


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize('C,O,H,W,s', [(96, 384, 8, 32, 2), (96, 384, 36, 64, 2), (96, 384, 45, 80, 2), (96, 384, 13, 37, 2),
                                       (96, 1152, 9, 33, 3)])
def test_bf16_block_fwd_bwd(orn, C, O, H, W, s):
    """conv3x3+PixelShuffle+SiLU fwd, and dbias / wgrad / dgrad, on bf16 MFMA vs the CPU oracle run on
    the SAME bf16-rounded inputs (isolates kernel correctness from input quantisation):
    exact integer-free check would hide layout bugs, so data are asymmetric random."""
    from oracle import cpu_ref
    L, P, st = orn._lib.lib(), orn._lib.ptr, orn._lib.stream
    gen = torch.Generator().manual_seed(C + O + H * W)
    x = _bf16_round(torch.randn(1, C, H, W, generator=gen))
    wf = _bf16_round(torch.randn(O, C, 3, 3, generator=gen) / math.sqrt(9 * C))
    bf = torch.randn(O, generator=gen) * 0.1
    Cn = O // (s * s)
    da = torch.randn(1, Cn, H * s, W * s, generator=gen)
    rx, rw, rb = (t.clone().requires_grad_(True) for t in (x, wf, bf))
    torch.set_num_threads(8)
    y = torch.nn.functional.conv2d(rx, rw, rb, padding=1)
    zr = torch.nn.functional.pixel_shuffle(y, s)
    ar = torch.nn.functional.silu(zr)
    nb = L.orn_conv3x3_ps_silu_bf16_ws_bytes(C, O, H, W, s)
    ws = torch.zeros(nb, dtype=torch.uint8, device='cuda')
    xd, wd, bd = x.cuda(), wf.cuda(), bf.cuda()
    z = torch.empty(1, Cn, H * s, W * s, device='cuda')
    a = torch.empty_like(z)
    orn._lib.check(L.orn_conv3x3_ps_silu_fwd_bf16(P(xd), P(wd), P(bd), C, O, H, W, s, P(z), P(a), P(ws), c_size_t(nb), st()))
    # outputs are stored as bf16: 2^-9 relative rounding on top of fp32 accumulation
    np.testing.assert_allclose(z.cpu().numpy(), zr.detach().numpy(), rtol=5e-3, atol=5e-3)
    np.testing.assert_allclose(a.cpu().numpy(), ar.detach().numpy(), rtol=5e-3, atol=5e-3)
    if C != 96:
        return
    # backward with the oracle's z and the same bf16-rounded dy the kernel sees
    zq = _bf16_round(zr.detach())
    dz = _bf16_round(da * (torch.sigmoid(zq) * (1 + zq * (1 - torch.sigmoid(zq)))))
    dy = torch.nn.functional.pixel_unshuffle(dz, s)
    gx, gw = torch.autograd.grad(y, (rx, rw), dy)
    gb = dy.sum(dim=(0, 2, 3))
    dx = torch.empty(1, C, H, W, device='cuda')
    dwf = torch.empty(O, C, 3, 3, device='cuda')
    dbf = torch.empty(O, device='cuda')
    zd, dad = zr.detach().cuda().contiguous(), da.cuda()       # keep alive: ptr() does not hold a reference
    orn._lib.check(L.orn_conv3x3_ps_silu_bwd_bf16(P(xd), P(wd), P(zd), P(dad), C, O, H, W, s,
                                                  P(dx), P(dwf), P(dbf), P(ws), c_size_t(nb), st()))
    torch.cuda.synchronize()
    sc = float(gw.abs().max())
    np.testing.assert_allclose(dwf.cpu().numpy(), gw.numpy(), rtol=2e-3, atol=2e-3 * sc)
    np.testing.assert_allclose(dbf.cpu().numpy(), gb.numpy(), rtol=2e-3, atol=2e-3 * float(gb.abs().max()))
    np.testing.assert_allclose(dx.cpu().numpy(), gx.numpy(), rtol=2e-3, atol=2e-3 * float(gx.abs().max()))
