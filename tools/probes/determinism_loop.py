"""Call the suite's own run-to-run determinism test repeatedly in one process.  usage: determinism_loop.py [rounds=15]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import orn_amd
from orn_amd import ops, model, utils, engine  # noqa
import test_gpu_bf16 as T
fn = T.test_engine_is_run_to_run_deterministic
fn = getattr(fn, '__wrapped__', fn)
bad = 0
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 15):
    for prec in ('fp32', 'bf16', 'fp16'):
        try:
            fn(orn_amd, prec)
        except AssertionError as e:
            bad += 1
            print('round', r, prec, 'DIFFERS', str(e)[:80], flush=True)
print('failures', bad)
