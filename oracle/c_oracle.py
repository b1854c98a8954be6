"""ctypes binding of oracle/merge_ref.c -- TEST INFRASTRUCTURE ONLY (see cpu_ref.py header)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liborn_oracle.so')
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, 'merge_ref.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def merge_fwd(w3x3, b3x3, w3x1, b3x1, w1x3, b1x3, w1, w2, w3):
    """-> (Wf [O,C,3,3], bf [O], T [O,C,3,3]); model.py:450-516 in the specified fmaf order."""
    w3x3, b3x3, w3x1, b3x1, w1x3, b1x3, w1, w2, w3 = map(_f32, (w3x3, b3x3, w3x1, b3x1, w1x3, b1x3, w1, w2, w3))
    O, C = w3x3.shape[0], w3x3.shape[1]
    T = np.empty((O, C, 3, 3), np.float32)
    wf = np.empty((O, C, 3, 3), np.float32)
    bf = np.empty((O,), np.float32)
    lib().orn_oracle_merge_fwd(_p(w3x3), _p(b3x3), _p(w3x1), _p(b3x1), _p(w1x3), _p(b1x3), _p(w1), _p(w2), _p(w3),
                               ctypes.c_int(C), ctypes.c_int(O), _p(T), _p(wf), _p(bf))
    return wf, bf, T


def merge_bwd(g, w1, w2, w3, T):
    """-> dict(dT, dW1, dW2, dW3) in the fmaf order documented in merge_ref.c."""
    g, w1, w2, w3, T = map(_f32, (g, w1, w2, w3, T))
    O, C = g.shape[0], g.shape[1]
    dT = np.empty((O, C, 3, 3), np.float32)
    dw1 = np.empty((2 * C, C, 1, 1), np.float32)
    dw2 = np.empty((O, 2 * C, 3, 3), np.float32)
    dw3 = np.empty((O, O, 1, 1), np.float32)
    lib().orn_oracle_merge_bwd(_p(g), _p(w1), _p(w2), _p(w3), _p(T), ctypes.c_int(C), ctypes.c_int(O),
                               _p(dT), _p(dw1), _p(dw2), _p(dw3))
    return dict(dT=dT, dW1=dw1, dW2=dw2, dW3=dw3)
