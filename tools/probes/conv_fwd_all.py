"""Launch the dominant kernel (16-bit conv forward) once per fast layer of BASELINE config 2 (L1 zero-padded to 96 channels, L2, L3, L4), after a
warm-up, for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE)."""
import sys, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from ctypes import c_void_p
import orn_amd
from orn_amd import _lib
lib = _lib.lib()
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
hdt = torch.float16 if prec == 'fp16' else torch.bfloat16
fwd = lib.orn_conv_nhwc_f16_fwd if prec == 'fp16' else lib.orn_conv_nhwc_bf16_fwd
st = _lib.stream()
keep = []
if len(sys.argv) > 2:                       # timing-only ablation flags of the conv kernel (results are wrong)
    from ctypes import c_int
    lib.orn_debug_set(c_int(int(sys.argv[2])))
for rep in range(2):
    for (H, W, Cr, last) in ((45, 80, 26, False), (90, 160, 96, False), (180, 320, 96, False), (360, 640, 96, True)):
        C, O, s = 96, 384, 2
        xpad = torch.zeros(H + 2, W + 2, C, device='cuda', dtype=hdt)
        xpad[1:-1, 1:-1, :Cr] = torch.randn(H, W, Cr, device='cuda').to(hdt)
        wb = torch.zeros(9, O, C, device='cuda', dtype=hdt)
        wb[:, :, :Cr] = (torch.randn(9, O, Cr, device='cuda') * 0.034).to(hdt)
        bp = torch.zeros(O, device='cuda')
        z = torch.empty(H * s, W * s, 96, device='cuda', dtype=hdt)
        apad = None if last else torch.zeros(H * s + 2, W * s + 2, 96, device='cuda', dtype=hdt)
        keep.append((xpad, wb, bp, z, apad))
        _lib.check(fwd(c_void_p(xpad.data_ptr()), c_void_p(wb.data_ptr()), _lib.ptr(bp), H, W, C, O, s, c_void_p(z.data_ptr()),
                       c_void_p(apad.data_ptr()) if apad is not None else None, st))
        torch.cuda.synchronize()
print('done')
