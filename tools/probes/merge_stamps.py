"""Phase stamps of the forward-merge GEMM (diagnostic build: ORN_BUILD_TAG=mstamp ORN_EXTRA_DEFS=-DORN_MERGE_STAMP, run with
ORN_LIB_PATH=.../liborn_mstamp.so): s_memtime ticks of work-group (0,0,0) of every problem shape, (a) inside real training steps
of the 720p engine (cold caches: the launch follows the previous step's Adam), (b) per-op calls back to back (warm).
Read the SHARES, not the lengths."""
import ctypes, os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import orn_amd
from orn_amd import _lib
import bench
lib = _lib.lib()
raw = ctypes.CDLL(os.environ['ORN_LIB_PATH'])


def dump(tag):
    out = (ctypes.c_ulonglong * 128)()
    raw.orn_debug_merge_stamps(out)
    for s in range(8):
        r = out[16 * s:16 * s + 16]
        if r[5]:
            print(f'{tag}: M={r[6]} N={r[7]} K={r[8]}: kernel {r[0]} = prologue {r[1]} + loop {r[2]} ({r[5]} chunks, {r[2] // max(r[5], 1)} per chunk) + epilogue {r[3]}')


eng = bench.make_engine(seed=1234, precision='fp16')
eng.set_schedule(bench.schedule(16))
eng.run(16, graph=False)
torch.cuda.synchronize()
dump('in-step (eager)')
dev, st, P = 'cuda', _lib.stream(), _lib.ptr
for (C, O) in [(26, 650), (96, 384)]:
    w3x3 = torch.randn(O, C, 3, 3, device=dev); w3x1 = torch.randn(O, C, 3, 1, device=dev); w1x3 = torch.randn(O, C, 1, 3, device=dev)
    b = [torch.randn(O, device=dev) for _ in range(3)]
    w1 = torch.randn(2 * C, C, device=dev); w2 = torch.randn(O, 2 * C, 3, 3, device=dev); w3 = torch.randn(O, O, device=dev)
    T = torch.empty(O, C, 3, 3, device=dev); wf = torch.empty(O, C, 3, 3, device=dev); bf = torch.empty(O, device=dev)
    for _ in range(5):
        _lib.check(lib.orn_erb_merge_fwd(P(w3x3), P(b[0]), P(w3x1), P(b[1]), P(w1x3), P(b[2]), P(w1), P(w2), P(w3), C, O, P(T), P(wf), P(bf), st), 'm')
    torch.cuda.synchronize()
dump('per-op, warm')
