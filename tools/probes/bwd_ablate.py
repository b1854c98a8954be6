"""(needs a library built with ORN_CONV_ABLATE=1: python -c "import os; os.environ['ORN_CONV_ABLATE']='1'; from orn_amd import _build; _build.build(force=True)")
Timing of the bf16 wgrad / dgrad kernels at the L4 shape (+ timing-only ablations)."""
import sys, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from ctypes import c_void_p, c_int
import orn_amd
from orn_amd import _lib
lib = _lib.lib()
H, W, C, O, s = 360, 640, 96, 384, 2
dev = 'cuda'
bf = torch.bfloat16
xpad = torch.zeros(H + 2, W + 2, C, device=dev, dtype=bf); xpad[1:-1, 1:-1] = torch.randn(H, W, C, device=dev).to(bf)
dypad = torch.zeros(H + 2, W + 2, O, device=dev, dtype=bf); dypad[1:-1, 1:-1] = (torch.randn(H, W, O, device=dev) * 1e-3).to(bf)
wd = (torch.randn(9, C, O, device=dev) * 0.03).to(bf)
zprev = torch.randn(H, W, C, device=dev).to(bf)
dyprev = torch.zeros(H // 2 + 2, W // 2 + 2, C * 4, device=dev, dtype=bf)
nb = lib.orn_wgrad_nhwc_bf16_ws_bytes(H, W, O)
slabs = torch.empty(nb // 4, device=dev)
dwf = torch.empty(O, C, 3, 3, device=dev); dbf = torch.empty(O, device=dev)
st = _lib.stream()
P = lambda t: c_void_p(t.data_ptr())
def wgrad(): _lib.check(lib.orn_wgrad_nhwc_bf16(P(xpad), P(dypad), H, W, C, O, s, P(slabs), P(dwf), P(dbf), st))
def dgrad(): _lib.check(lib.orn_dgrad_nhwc_bf16(P(dypad), P(wd), H, W, O, C, P(zprev), P(dyprev), 2, st))
def t(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
names = {0: 'baseline', 1: 'no weight restage', 2: 'no patch stage', 4: 'no stores', 3: 'no global loads', 7: 'mfma+lds+barriers', 15: 'mfma+lds', 31: 'mfma only'}
for rnd in range(3):
    for f in (0, 2, 1, 4, 7, 31):
        lib.orn_debug_set(c_int(f))
        ms = t(dgrad); print(f'round {rnd} dgrad flags {f} ({names[f]}): {ms*1e3:.1f} us {152.9e9/ms/1e9:.0f} TF')
lib.orn_debug_set(c_int(0))
