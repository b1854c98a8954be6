"""(needs a library built with ORN_CONV_ABLATE=1: python -c "import os; os.environ['ORN_CONV_ABLATE']='1'; from orn_amd import _build; _build.build(force=True)")
Timing-only ablation of the bf16 conv kernel at the L4 shape (results are WRONG under flags)."""
import sys, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from ctypes import c_void_p, c_int
import orn_amd
from orn_amd import _lib
lib = _lib.lib()
H, W, C, O, s = 360, 640, 96, 384, 2
dev = 'cuda'
xpad = torch.zeros(H + 2, W + 2, C, device=dev, dtype=torch.bfloat16)
xpad[1:-1, 1:-1] = torch.randn(H, W, C, device=dev).to(torch.bfloat16)
wb = (torch.randn(9, O, C, device=dev) * (1.0 / (9 * C) ** 0.5)).to(torch.bfloat16)
bp = torch.zeros(O, device=dev)
z = torch.empty(H * s, W * s, O // 4, device=dev, dtype=torch.bfloat16)
st = _lib.stream()
def run():
    _lib.check(lib.orn_conv_nhwc_bf16_fwd(c_void_p(xpad.data_ptr()), c_void_p(wb.data_ptr()), _lib.ptr(bp), H, W, C, O, s, c_void_p(z.data_ptr()), None, st))
names = {0: 'baseline', 7: 'mfma+lds+barriers', 15: 'mfma+lds (no barriers)', 23: 'mfma+barriers (no lds reads)', 31: 'mfma only'}
for rnd in range(3):
    for f in (31, 0):
        lib.orn_debug_set(c_int(f))
        for _ in range(2): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f'round {rnd} flags {f} ({names[f]}): {ms*1e3:.1f} us  {152.9e9/ms/1e9:.0f} TF')
lib.orn_debug_set(c_int(0))
