"""MI355X-native Online-RepNeRV training hot path.

Host-side mirror of the reference's Python surface for this path (model.py / utils.py /
main_train.py of maoqingyu1996/Boosting-Neural-Video-Representation-via-Online-Structural-
Reparameteration) on top of liborn.so, the hand-written gfx950 HIP library behind include/orn.h.
There is no CPU or PyTorch fallback: every op raises if the library is missing or the tensors are
not on the GPU.
"""
from . import _lib  # noqa: F401
from ._lib import OrnError, lib_path  # noqa: F401

__all__ = ['_lib', 'OrnError', 'lib_path']
__version__ = '0.1.0'
