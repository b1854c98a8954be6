"""Run the fused dgrad (orn_dgrad_nhwc_f16) on fixed random data and save / compare dyprev.  usage: dgrad_dump.py save|cmp <file> [H W]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from ctypes import c_void_p
import orn_amd
from orn_amd import _lib
lib = _lib.lib()
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (64, 512)
C, O = 96, 384
torch.manual_seed(0)
dev = 'cuda'
dypad = torch.zeros(H + 2, W + 2, O, device=dev, dtype=torch.float16)
dypad[1:-1, 1:-1] = torch.randn(H, W, O, device=dev).half()
wd = (torch.randn(9, C, O, device=dev) / (9 * O) ** 0.5).half()
wd = torch.cat([wd.flatten(), torch.zeros(96 * 96 * 9, device=dev, dtype=torch.float16)])
zprev = torch.randn(H, W, C, device=dev).half()
dyprev = torch.zeros(H // 2 + 2, W // 2 + 2, C * 4, device=dev, dtype=torch.float16)
P = lambda t: c_void_p(t.data_ptr())
_lib.check(lib.orn_dgrad_nhwc_f16(P(dypad), P(wd), H, W, O, C, P(zprev), P(dyprev), 2, _lib.stream()))
torch.cuda.synchronize()
if sys.argv[1] == 'save':
    torch.save(dyprev.cpu(), sys.argv[2])
else:
    ref = torch.load(sys.argv[2]).float()
    d = (dyprev.cpu().float() - ref)
    print('max abs diff', float(d.abs().max()), 'ref absmax', float(ref.abs().max()), 'mismatching', float((d.abs() > 1e-2).float().mean()))
    # dyprev [ph+1][pw+1][sub*96 + c]: which (sub, channel group, pixel) differ
    dd = (d.abs() > 1e-2)[1:-1, 1:-1].view(H // 2, W // 2, 4, 96)
    print('by sub-position:', [round(float(dd[:, :, k].float().mean()), 3) for k in range(4)])
    print('by channel (groups of 4):', [round(float(dd[..., k:k + 4].float().mean()), 2) for k in range(0, 96, 4)])
    print('by ph mod 4:', [round(float(dd[k::4].float().mean()), 2) for k in range(4)], 'by pw mod 16:', [round(float(dd[:, k::16].float().mean()), 2) for k in range(16)])
