"""Per-layer timing of the ERB merge forward / backward ops (ungrouped launches) at the 720p shapes."""
import sys, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import orn_amd
from orn_amd import _lib
from ctypes import c_size_t
lib = _lib.lib()
dev = 'cuda'
st = _lib.stream()
P = _lib.ptr


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (C, O, flag) in [(26, 650, 0), (26, 384, 0), (96, 384, 0)]:
    w3x3 = torch.randn(O, C, 3, 3, device=dev); w3x1 = torch.randn(O, C, 3, 1, device=dev); w1x3 = torch.randn(O, C, 1, 3, device=dev)
    b = [torch.randn(O, device=dev) for _ in range(3)]
    w1 = torch.randn(2 * C, C, device=dev); w2 = torch.randn(O, 2 * C, 3, 3, device=dev); w3 = torch.randn(O, O, device=dev)
    T = torch.empty(O, C, 3, 3, device=dev); wf = torch.empty(O, C, 3, 3, device=dev); bf = torch.empty(O, device=dev)
    f = lambda: _lib.check(lib.orn_erb_merge_fwd(P(w3x3), P(b[0]), P(w3x1), P(b[1]), P(w1x3), P(b[2]), P(w1), P(w2), P(w3), C, O, P(T), P(wf), P(bf), st), 'm')
    print(f'flag {flag}: C={C} O={O}: merge fwd (T + S + bias, 3 launches) {timeit(f):.1f} us')
    if len(sys.argv) > 1 and sys.argv[1] == 'fwd':
        continue
    g = torch.randn(O, C, 3, 3, device=dev) * 1e-6; dbf = torch.randn(O, device=dev)
    outs = [torch.empty_like(x) for x in (w3x3, b[0], w3x1, b[1], w1x3, b[2], w1, w2, w3)]
    nb = lib.orn_erb_merge_bwd_ws_bytes(C, O)
    ws = torch.empty(nb // 4 + 64, device=dev)
    h = lambda: _lib.check(lib.orn_erb_merge_bwd(P(g), P(dbf), P(w1), P(w2), P(w3), P(T), C, O, *[P(o) for o in outs], P(ws), c_size_t(nb), st), 'b')
    print(f'flag {flag}: C={C} O={O}: merge bwd (slices + 4 GEMMs + reduce, 6 launches) {timeit(h):.1f} us')
