"""Where is the 16-bit block forward wrong?  Error of orn_conv3x3_ps_silu_fwd_f16 vs torch on the GPU, broken down by output
sub-position (si, sj), channel, tile row and tile column.  usage: fwd_errmap.py [H=64] [W=512] [s=2]"""
import os, sys, math, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from ctypes import c_size_t
import orn_amd
from orn_amd import _lib
L, P, st = _lib.lib(), _lib.ptr, _lib.stream
H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = int(sys.argv[2]) if len(sys.argv) > 2 else 512
s = int(sys.argv[3]) if len(sys.argv) > 3 else 2
C, O = 96, 96 * s * s
g = torch.Generator().manual_seed(3)
x = torch.randn(1, C, H, W, generator=g).half().float().cuda()
wf = (torch.randn(O, C, 3, 3, generator=g) / math.sqrt(9 * C)).half().float().cuda()
bf = (torch.randn(O, generator=g) * 0.1).cuda()
if os.environ.get('ERR_CHUNK'):          # keep only one 32-channel chunk / one tap of the weights
    q = int(os.environ['ERR_CHUNK']); m = torch.zeros_like(wf); m[:, 32 * q:32 * q + 32] = 1; wf = wf * m
if os.environ.get('ERR_TAP'):
    tp = int(os.environ['ERR_TAP']); m = torch.zeros_like(wf); m[:, :, tp // 3, tp % 3] = 1; wf = wf * m
zr = torch.nn.functional.pixel_shuffle(torch.nn.functional.conv2d(x, wf, bf, padding=1), s)
nb = L.orn_conv3x3_ps_silu_bf16_ws_bytes(C, O, H, W, s)
ws = torch.zeros(nb, dtype=torch.uint8, device='cuda')
z = torch.empty(1, C, H * s, W * s, device='cuda'); a = torch.empty_like(z)
_lib.check(L.orn_conv3x3_ps_silu_fwd_f16(P(x), P(wf), P(bf), C, O, H, W, s, P(z), P(a), P(ws), c_size_t(nb), st()))
torch.cuda.synchronize()
bad = (~((z - zr).abs() <= 2e-2))[0]                      # [Cn][H*s][W*s]
print('bad fraction', float(bad.float().mean()), 'nan count', int(torch.isnan(z).sum()), 'bad count', int(bad.sum()))
idx = bad.nonzero()[:12].tolist()
print('first bad (n, oh, ow):', idx, [float(z[0][tuple(i)]) for i in idx[:6]], [float(zr[0][tuple(i)]) for i in idx[:6]])
b5 = bad.view(C, H, s, W, s)                          # n, h, si, w, sj
print('by (si,sj):', [[round(float(b5[:, :, i, :, j].float().mean()), 3) for j in range(s)] for i in range(s)])
print('by channel n (groups of 8):', [round(float(b5[k:k + 8].float().mean()), 2) for k in range(0, C, 8)])
print('by h mod 8:', [round(float(b5[:, k::8].float().mean()), 2) for k in range(8)])
print('by w mod 32 (groups of 4):', [round(float(torch.stack([b5[:, :, :, k + d::32] for d in range(4)]).float().mean()), 2) for k in range(0, 32, 4)])
print('by tile row:', [round(float(b5[:, k:k + 8].float().mean()), 2) for k in range(0, H, 8)])
print('by tile col:', [round(float(b5[:, :, :, k:k + 32].float().mean()), 2) for k in range(0, W, 32)])
