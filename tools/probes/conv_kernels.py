"""Time the three 16-bit conv kernels (forward of the last block, dgrad with the fused epilogue, wgrad) at a 720p layer shape
through the C ABI, on random data.  usage: conv_kernels.py [which=fwd,dgrad,wgrad] [iters] [layer=4|3]   (ORN_LIB_PATH selects
another build for A/B runs; also used under rocprofv3 --pmc)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from ctypes import c_void_p
import orn_amd
from orn_amd import _lib
lib = _lib.lib()
which = (sys.argv[1] if len(sys.argv) > 1 else 'fwd,dgrad,wgrad').split(',')
half = os.environ.get('ORN_HALF', 'bf16')                 # element type of the buffers / kernel build: bf16 | fp16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
layer = int(sys.argv[3]) if len(sys.argv) > 3 else 4
H, W = {4: (360, 640), 3: (180, 320), 2: (90, 160)}[layer]
C, O, s = 96, 384, 2
dev = 'cuda'
torch.manual_seed(0)
bf = torch.bfloat16 if half == 'bf16' else torch.float16
FWD = lib.orn_conv_nhwc_bf16_fwd if half == 'bf16' else lib.orn_conv_nhwc_f16_fwd
DGRAD = lib.orn_dgrad_nhwc_bf16 if half == 'bf16' else lib.orn_dgrad_nhwc_f16
WGRAD = lib.orn_wgrad_nhwc_bf16 if half == 'bf16' else lib.orn_wgrad_nhwc_f16
xpad = torch.zeros(H + 2, W + 2, C, device=dev, dtype=bf)
xpad[1:-1, 1:-1] = torch.randn(H, W, C, device=dev).to(bf)
wb = (torch.randn(9, O, C, device=dev) * (1.0 / (9 * C) ** 0.5)).to(bf)
wd = (torch.randn(9, C, O, device=dev) * (1.0 / (9 * O) ** 0.5)).to(bf)
bp = torch.randn(O, device=dev)
z = torch.empty(H * s, W * s, O // 4, device=dev, dtype=bf)
dypad = torch.zeros(H + 2, W + 2, O, device=dev, dtype=bf)
dypad[1:-1, 1:-1] = torch.randn(H, W, O, device=dev).to(bf)
zprev = torch.randn(H, W, C, device=dev).to(bf)
dyprev = torch.zeros(H // 2 + 2, W // 2 + 2, C * 4, device=dev, dtype=bf)
slabs = torch.empty(lib.orn_wgrad_nhwc_bf16_ws_bytes(H, W, O) // 4, device=dev)
dwf = torch.empty(O, C, 3, 3, device=dev)
dbf = torch.empty(O, device=dev)
st = _lib.stream()
if os.environ.get('ORN_DBG'):                             # ablation flags of a -DORN_CONV_ABLATE build
    lib.orn_debug_set(int(os.environ['ORN_DBG']))
P = lambda t: c_void_p(t.data_ptr())
fl = 2.0 * 9 * C * O * H * W


def fwd():
    _lib.check(FWD(P(xpad), P(wb), _lib.ptr(bp), H, W, C, O, s, P(z), None, st))


def dgrad():
    _lib.check(DGRAD(P(dypad), P(wd), H, W, O, C, P(zprev), P(dyprev), 2, st))


def wgrad():
    _lib.check(WGRAD(P(xpad), P(dypad), H, W, C, O, s, P(slabs), P(dwf), P(dbf), st))


for name in which:
    f = {'fwd': fwd, 'dgrad': dgrad, 'wgrad': wgrad}[name]
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    best, tot = 1e9, 0.0
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / iters
        best = min(best, ms); tot += ms
    print(f'{name} L{layer}: best {best*1e3:.1f} us mean {tot/3*1e3:.1f} us  {fl/best/1e9:.0f} TF (best)', flush=True)
