"""(needs a library built with ORN_CONV_ABLATE=1, see bwd_ablate.py; pass it through ORN_LIB_PATH)
Timing of the 16-bit wgrad kernel + reduce at the four fast-layer shapes, with timing-only ablations:
1 no LDS-DMA (operands stay zero), 4 no slab stores, 32 no dbias MFMA."""
import sys, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from ctypes import c_void_p, c_int
import orn_amd
from orn_amd import _lib
lib = _lib.lib()
dev = 'cuda'
hf = torch.bfloat16
st = _lib.stream()
P = lambda t: c_void_p(t.data_ptr())
def t(fn, n=20):
    if len(sys.argv) > 1: n = 2
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
for (H, W) in ((360, 640), (180, 320), (90, 160), (45, 80)):
    C, O, s = 96, 384, 2
    xpad = torch.zeros(H + 2, W + 2, C, device=dev, dtype=hf); xpad[1:-1, 1:-1] = torch.randn(H, W, C, device=dev).to(hf)
    dypad = torch.zeros(H + 2, W + 2, O, device=dev, dtype=hf); dypad[1:-1, 1:-1] = torch.randn(H, W, O, device=dev).to(hf)
    nb = lib.orn_wgrad_nhwc_bf16_ws_bytes(H, W, O)
    slabs = torch.empty(nb // 4, device=dev)
    dwf = torch.empty(O, C, 3, 3, device=dev); dbf = torch.empty(O, device=dev)
    def wgrad(): _lib.check(lib.orn_wgrad_nhwc_bf16(P(xpad), P(dypad), H, W, C, O, s, P(slabs), P(dwf), P(dbf), st))
    gf = 2.0 * H * W * O * C * 9
    for f in ((0,) if len(sys.argv) > 1 else (0, 1, 4, 5, 32)):
        lib.orn_debug_set(c_int(f))
        ms = t(wgrad)
        print(f'{H}x{W} flags {f}: {ms*1e3:.1f} us (wgrad + reduce) {gf/ms/1e9:.0f} TF', flush=True)
    lib.orn_debug_set(c_int(0))
