"""Run the bf16 conv kernels at the L4 shape a few times (for rocprofv3 --pmc passes)."""
import sys, torch
sys.path.insert(0, '.')
from ctypes import c_void_p
import orn_amd
from orn_amd import _lib
lib = _lib.lib()
H, W, C, O, s = 360, 640, 96, 384, 2
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = 'cuda'
xpad = torch.zeros(H + 2, W + 2, C, device=dev, dtype=torch.bfloat16)
xpad[1:-1, 1:-1] = torch.randn(H, W, C, device=dev).to(torch.bfloat16)
wb = (torch.randn(9, O, C, device=dev) * (1.0 / (9 * C) ** 0.5)).to(torch.bfloat16)
bp = torch.zeros(O, device=dev)
z = torch.empty(H * s, W * s, O // 4, device=dev, dtype=torch.bfloat16)
st = _lib.stream()
for _ in range(iters):
    _lib.check(lib.orn_conv_nhwc_bf16_fwd(c_void_p(xpad.data_ptr()), c_void_p(wb.data_ptr()), _lib.ptr(bp), H, W, C, O, s, c_void_p(z.data_ptr()), None, st))
torch.cuda.synchronize()
print('done')
