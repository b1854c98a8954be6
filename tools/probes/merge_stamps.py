"""Phase stamps of the forward-merge GEMM (diagnostic build: ORN_BUILD_TAG=mstamp ORN_EXTRA_DEFS=-DORN_MERGE_STAMP, run with
ORN_LIB_PATH=.../liborn_mstamp.so): s_memtime sums of work-group (0,0,0) per problem shape.  Read the SHARES, not the lengths."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import orn_amd
from orn_amd import _lib
lib = _lib.lib()
raw = ctypes.CDLL(os.environ['ORN_LIB_PATH'])
dev = 'cuda'
st = _lib.stream()
P = _lib.ptr
names = ['kernel', 'prologue', 'K loop', 'epilogue', 'sum lstore(+load wait)', 'sum gload issue', '-', 'sum reads+MFMA+barrier', 'chunks']
for (C, O) in [(26, 650), (26, 384), (96, 384)]:
    w3x3 = torch.randn(O, C, 3, 3, device=dev); w3x1 = torch.randn(O, C, 3, 1, device=dev); w1x3 = torch.randn(O, C, 1, 3, device=dev)
    b = [torch.randn(O, device=dev) for _ in range(3)]
    w1 = torch.randn(2 * C, C, device=dev); w2 = torch.randn(O, 2 * C, 3, 3, device=dev); w3 = torch.randn(O, O, device=dev)
    T = torch.empty(O, C, 3, 3, device=dev); wf = torch.empty(O, C, 3, 3, device=dev); bf = torch.empty(O, device=dev)
    for _ in range(5):
        _lib.check(lib.orn_erb_merge_fwd(P(w3x3), P(b[0]), P(w3x1), P(b[1]), P(w1x3), P(b[2]), P(w1), P(w2), P(w3), C, O, P(T), P(wf), P(bf), st), 'm')
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    raw.orn_debug_merge_stamps(out)          # the LAST GEMM launch of the op is the S product
    print(f'C={C} O={O} S product: ' + ', '.join(f'{n} {int(v)}' for n, v in zip(names, out) if n != '-'))
