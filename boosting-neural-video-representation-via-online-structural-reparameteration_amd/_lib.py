"""ctypes binding of liborn.so (include/orn.h).  Fails loudly when the library is missing."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
ORN_MAX_LAYERS = 8
LOSS_TYPES = {'L2': 0, 'L1': 1, 'Fusion6': 2}


class OrnError(RuntimeError):
    pass


def lib_path() -> str:
    # ORN_LIB_PATH: A/B timing of two builds on one GPU box (tools/probes); the default is the in-tree library
    return os.environ.get('ORN_LIB_PATH') or os.path.join(HERE, 'liborn.so')


class LayerDesc(ctypes.Structure):
    _fields_ = [('C', c_int32), ('O', c_int32), ('s', c_int32), ('H', c_int32), ('W', c_int32),
                ('w3x3', c_int64), ('b3x3', c_int64), ('w3x1', c_int64), ('b3x1', c_int64),
                ('w1x3', c_int64), ('b1x3', c_int64), ('w1', c_int64), ('w2', c_int64), ('w3', c_int64)]


class EngineDesc(ctypes.Structure):
    _fields_ = [('n_layers', c_int32), ('erb', c_int32), ('embed_len', c_int32), ('stem_dim', c_int32),
                ('fc_h', c_int32), ('fc_w', c_int32), ('fc_dim', c_int32), ('sigmoid', c_int32),
                ('loss_type', c_int32), ('precision', c_int32),
                ('beta1', c_double), ('beta2', c_double), ('eps', c_double),
                ('stem_w0', c_int64), ('stem_b0', c_int64), ('stem_w1', c_int64), ('stem_b1', c_int64),
                ('head_w', c_int64), ('head_b', c_int64), ('n_params', c_int64),
                ('layer', LayerDesc * ORN_MAX_LAYERS)]


P = c_void_p
_SIGS = {
    'orn_version': (c_int, []),
    'orn_last_error': (c_int, [c_char_p, c_size_t]),
    'orn_pe_fwd': (c_int, [P, c_int, P, c_int, P, P]),
    'orn_stem_fwd': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, P]),
    'orn_stem_bwd': (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, P]),
    'orn_erb_merge_fwd': (c_int, [P] * 9 + [c_int, c_int, P, P, P, P]),
    'orn_erb_merge_bwd_ws_bytes': (c_size_t, [c_int, c_int]),
    'orn_erb_merge_bwd': (c_int, [P] * 6 + [c_int, c_int] + [P] * 9 + [P, c_size_t, P]),
    'orn_conv3x3_ps_silu_fwd': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    'orn_conv3x3_ps_silu_bwd_ws_bytes': (c_size_t, [c_int] * 5),
    'orn_conv3x3_ps_silu_bwd': (c_int, [P, P, P, P] + [c_int] * 6 + [P, P, P, P, c_size_t, P]),
    'orn_conv3x3_ps_silu_bf16_ws_bytes': (c_size_t, [c_int] * 5),
    'orn_conv3x3_ps_silu_fwd_bf16': (c_int, [P, P, P] + [c_int] * 5 + [P, P, P, c_size_t, P]),
    'orn_conv3x3_ps_silu_bwd_bf16': (c_int, [P, P, P, P] + [c_int] * 5 + [P, P, P, P, c_size_t, P]),
    'orn_conv_nhwc_bf16_fwd': (c_int, [P, P, P] + [c_int] * 5 + [P, P, P]),
    'orn_conv_nhwc_f16_fwd': (c_int, [P, P, P] + [c_int] * 5 + [P, P, P]),
    'orn_dgrad_nhwc_bf16': (c_int, [P, P] + [c_int] * 4 + [P, P, c_int, P]),
    'orn_wgrad_nhwc_bf16_ws_bytes': (c_size_t, [c_int] * 3),
    'orn_wgrad_nhwc_bf16': (c_int, [P, P] + [c_int] * 5 + [P, P, P, P]),
    'orn_dgrad_nhwc_f16': (c_int, [P, P] + [c_int] * 4 + [P, P, c_int, P]),
    'orn_wgrad_nhwc_f16': (c_int, [P, P] + [c_int] * 5 + [P, P, P, P]),
    'orn_conv3x3_ps_silu_fwd_f16': (c_int, [P, P, P] + [c_int] * 5 + [P, P, P, c_size_t, P]),
    'orn_conv3x3_ps_silu_bwd_f16': (c_int, [P, P, P, P] + [c_int] * 5 + [P, P, P, P, c_size_t, P]),
    'orn_head_fwd': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    'orn_head_bwd_ws_bytes': (c_size_t, [c_int] * 4),
    'orn_head_bwd': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_size_t, P]),
    'orn_loss_ws_bytes': (c_size_t, [c_int] * 4),
    'orn_loss_fwd_bwd': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_float, P, P, P, c_size_t, P]),
    'orn_loss_target_stats_bytes': (c_size_t, [c_int] * 4),
    'orn_loss_target_stats': (c_int, [P, c_int, c_int, c_int, c_int, P, P]),
    'orn_msssim_ws_bytes': (c_size_t, [c_int] * 4),
    'orn_msssim': (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, c_size_t, P]),
    'orn_adam_step': (c_int, [P, P, P, P, c_size_t, c_double, c_double, c_double, c_double, c_int, P]),
    'orn_engine_ws_bytes': (c_size_t, [POINTER(EngineDesc)]),
    'orn_engine_create': (c_int, [POINTER(EngineDesc), P, P, P, P, P, c_size_t, POINTER(c_void_p)]),
    'orn_engine_destroy': (None, [P]),
    'orn_engine_decode': (c_int, [P, P, P, P]),
    'orn_engine_train_step': (c_int, [P, P, P, P, P, P, c_int32, P]),
    'orn_engine_train_steps_graph': (c_int, [P, P, P, P, P, P, c_int32, c_int32, P]),
    'orn_engine_train_steps': (c_int, [P, P, P, P, P, P, c_int32, c_int32, P]),
    'orn_engine_profile_step': (c_int, [P, P, P, P, P, P, c_int32, P, P]),
    'orn_engine_set_grad_mask': (c_int, [P, P]),
    'orn_engine_set_target_stats': (c_int, [P, P]),
    'orn_engine_fused_kernel': (c_int, [P, c_int, POINTER(c_void_p), POINTER(c_void_p)]),
    'orn_engine_scale_state': (c_int, [P, P]),
    'orn_engine_set_grad_scale': (c_int, [P, c_float, c_float]),
}
EXPORTS = tuple(_SIGS.keys())
# probe-only entry points (include/orn_debug.h): resolved if present, never required
_DEBUG_SIGS = {'orn_debug_set': (None, [c_int]), 'orn_debug_set_stamps': (None, [c_void_p])}

_lib = None


def lib():
    """The loaded library.  Raises OrnError (never falls back) if liborn.so has not been built."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise OrnError(f'{path} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           '(hipcc --offload-arch=gfx950).  There is no CPU fallback.')
        L = ctypes.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)          # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in _DEBUG_SIGS.items():
            fn = getattr(L, name, None)
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    buf = ctypes.create_string_buffer(512)
    lib().orn_last_error(buf, 512)
    return buf.value.decode('utf-8', 'replace')


def check(rc: int, what: str = ''):
    if rc != 0:
        raise OrnError(f'{what or "liborn"} failed (rc={rc}): {last_error()}')


def ptr(t):
    """Raw device pointer of a contiguous CUDA (HIP) fp32 tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise OrnError('liborn ops need tensors on the GPU (there is no CPU path)')
    if not t.is_contiguous():
        raise OrnError('liborn ops need contiguous tensors')
    return c_void_p(t.data_ptr())


def stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
