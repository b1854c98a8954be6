"""(needs a library built with ORN_CONV_ABLATE=1, passed through ORN_LIB_PATH) wall-clock stamps (100 MHz) of the phases of the
fused first-block kernels, work-group 0 / thread 0, inside the replayed 720p training step."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from orn_amd import _lib
eng = bench.make_engine(1234, 'fp16')
eng.set_schedule(bench.schedule(64))
eng.run(32)
eng.run(32)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
fn = _lib.lib().orn_stage0_diag              # same handle as the engine's (ctypes.CDLL caches by path)
fn.restype = ctypes.c_int
assert fn(buf) == 0
v = list(buf)
f = [(v[i + 1] - v[i]) * 10 for i in range(0, 4)]
b = [(v[i + 1] - v[i]) * 10 for i in range(8, 13)]
print('fwd ns: loads+fill %d, barrier %d, gemm %d, epilogue %d' % tuple(f))
print('bwd ns: loads+fill %d, barrier %d, dbias %d, wgrad %d, dgrad+store %d' % tuple(b))
