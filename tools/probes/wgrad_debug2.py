import sys, math, torch
sys.path.insert(0, '.')
from ctypes import c_size_t
import orn_amd
L, P, st = orn_amd._lib.lib(), orn_amd._lib.ptr, orn_amd._lib.stream
C, O, H, W, s = 96, 384, 8, 32, 2
Cn = O // 4
g = torch.Generator().manual_seed(0)
x = torch.randn(1, C, H, W, generator=g).to(torch.bfloat16).float()
wf = torch.zeros(O, C, 3, 3)
nb = L.orn_conv3x3_ps_silu_bf16_ws_bytes(C, O, H, W, s)
ws = torch.zeros(nb, dtype=torch.uint8, device='cuda')
xd, wd = x.cuda(), wf.cuda()
xp = torch.nn.functional.pad(x[0], (1, 1, 1, 1))
for (h, w) in [(1, 3), (2, 20), (5, 9)]:
  res = []
  for op in [0, 5, 15, 16, 31, 32, 47, 63, 64, 100, 127, 128, 160, 255, 256, 300, 383]:
    ij, n = divmod(op, Cn)
    oh, ow = h * 2 + ij // 2, w * 2 + ij % 2
    z = torch.zeros(1, Cn, H * s, W * s, device='cuda')
    da = torch.zeros_like(z)
    da[0, n, oh, ow] = 2.0
    dx = torch.empty(1, C, H, W, device='cuda'); dwf = torch.empty(O, C, 3, 3, device='cuda'); dbf = torch.empty(O, device='cuda')
    orn_amd._lib.check(L.orn_conv3x3_ps_silu_bwd_bf16(P(xd), P(wd), P(z), P(da), C, O, H, W, s, P(dx), P(dwf), P(dbf), P(ws), c_size_t(nb), st()))
    torch.cuda.synchronize()
    o = n * 4 + ij
    exp = xp[:, h:h + 3, w:w + 3]
    got = dwf.cpu()[o]
    ratio = (got * exp).sum() / (exp * exp).sum()
    resid = (got - ratio * exp).abs().max().item()
    others = dwf.cpu().abs().sum(dim=(1, 2, 3)); others[o] = 0
    res.append((op, round(ratio.item(), 4), round(resid, 4), round(dbf[o].item(), 3), int((others > 0).sum())))
  print((h, w), res)
