"""TrainEngine: the reference's per-frame training step (main_train.py:229-254) driven through
`orn_engine_*` (include/orn.h).

PyTorch is the allocator only: the module's parameters are re-homed as views into one flat fp32
arena (so `state_dict()`, checkpoints and the eager autograd path keep working on the same
memory), gradients and Adam state get arenas of the same shape, the video is resident in HBM, and
every step reads its frame index / LR / step count from a device-side schedule, so whole epochs are
enqueued without a host sync -- as plain stream launches pipelined over the engine's second stream
(`run()`'s default, orn_engine_train_steps) or as hipGraph replays of the serial step (`graph=True`).
"""
import ctypes
from ctypes import byref, c_int32, c_size_t, c_void_p
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops
from ._lib import EngineDesc, OrnError, check, lib

_ALIGN = 64          # floats (256 B): every tensor starts on a 256-byte boundary inside the arena
_ERB_FIELDS = {
    'rbr_3x3_branch.weight': 'w3x3', 'rbr_3x3_branch.bias': 'b3x3',
    'rbr_3x1_branch.weight': 'w3x1', 'rbr_3x1_branch.bias': 'b3x1',
    'rbr_1x3_branch.weight': 'w1x3', 'rbr_1x3_branch.bias': 'b1x3',
    'rbr_1x1_3x3_1x1_branch_1x1_1.weight': 'w1', 'rbr_1x1_3x3_1x1_branch_3x3.weight': 'w2',
    'rbr_1x1_3x3_1x1_branch_1x1_2.weight': 'w3',
}
_SINGLE_FIELDS = {'branch.weight': 'w3x3', 'branch.bias': 'b3x3', 'rbr_reparam.weight': 'w3x3', 'rbr_reparam.bias': 'b3x3'}


def arena_layout(named_shapes: Sequence[Tuple[str, Tuple[int, ...]]]):
    """name -> (offset, numel) with every tensor aligned to 256 B; returns (layout, total_floats)."""
    off = 0
    out = {}
    for name, shape in named_shapes:
        n = int(np.prod(shape)) if len(shape) else 1
        out[name] = (off, n)
        off += (n + _ALIGN - 1) // _ALIGN * _ALIGN
    return out, off


def build_desc(model, layout, total, loss_type='Fusion6', beta1=0.5, beta2=0.999, eps=1e-8, precision=0) -> EngineDesc:
    """Describe `model` (our Generator mirror) to the native engine."""
    d = EngineDesc()
    n_layers = len(model.layers)
    if n_layers > _lib.ORN_MAX_LAYERS:
        raise OrnError(f'{n_layers} layers > ORN_MAX_LAYERS')
    blk0 = model.layers[0]
    erb = (not blk0.deploy) and blk0.branch_type == 'ERB'
    d.n_layers, d.erb = n_layers, int(erb)
    d.embed_len, d.stem_dim = model.stem[0].in_features, model.stem[0].out_features
    d.fc_h, d.fc_w, d.fc_dim = model.fc_h, model.fc_w, model.fc_dim
    d.sigmoid = int(bool(model.sigmoid))
    d.loss_type = _lib.LOSS_TYPES[loss_type]
    d.precision = precision
    d.beta1, d.beta2, d.eps = beta1, beta2, eps
    d.stem_w0, d.stem_b0 = layout['stem.0.weight'][0], layout['stem.0.bias'][0]
    d.stem_w1, d.stem_b1 = layout['stem.2.weight'][0], layout['stem.2.bias'][0]
    d.head_w, d.head_b = layout[f'head_layers.{n_layers - 1}.weight'][0], layout[f'head_layers.{n_layers - 1}.bias'][0]
    d.n_params = total
    H, W = model.fc_h, model.fc_w
    for i, blk in enumerate(model.layers):
        L = d.layer[i]
        L.C, L.O, L.s, L.H, L.W = blk.ngf, blk.out_channels, blk.stride, H, W
        for f in ('w3x3', 'b3x3', 'w3x1', 'b3x1', 'w1x3', 'b1x3', 'w1', 'w2', 'w3'):
            setattr(L, f, -1)
        fields = _ERB_FIELDS if erb else _SINGLE_FIELDS
        for key, f in fields.items():
            name = f'layers.{i}.{key}'
            if name in layout:
                setattr(L, f, layout[name][0])
        H, W = H * blk.stride, W * blk.stride
    return d


def make_schedule(entries: Sequence[Tuple[int, int, float]]) -> np.ndarray:
    """[(frame, step, lr)] -> int32 [n,4] array with the orn_step_sched memory layout."""
    arr = np.zeros((len(entries), 4), dtype=np.int32)
    for i, (frame, step, lr) in enumerate(entries):
        arr[i, 0] = frame
        arr[i, 1] = step
        arr[i, 2] = np.float32(lr).view(np.int32)
    return arr


class TrainEngine:
    """Native training engine for one video (one process / one GPU per video)."""

    def __init__(self, model, loss_type: str = 'Fusion6', beta: float = 0.5, precision: str = 'fp32',
                 device: Optional[torch.device] = None, n_slots: int = 4096, target_cache: bool = True):
        if not torch.cuda.is_available():
            raise OrnError('TrainEngine needs a GPU: there is no CPU path')
        self.device = torch.device(device or f'cuda:{torch.cuda.current_device()}')
        self.model = model
        named = [(k, tuple(p.shape)) for k, p in model.named_parameters()]
        self.layout, self.n_params = arena_layout(named)
        dev = self.device
        self.params = torch.zeros(self.n_params, device=dev)
        self.grads = torch.zeros(self.n_params, device=dev)
        self.adam_m = torch.zeros(self.n_params, device=dev)
        self.adam_v = torch.zeros(self.n_params, device=dev)
        with torch.no_grad():
            for k, p in model.named_parameters():
                off, n = self.layout[k]
                view = self.params[off:off + n].view(p.shape)
                view.copy_(p.detach().to(dev))
                p.data = view                                   # parameters now live in the arena
                p.grad = self.grads[off:off + n].view(p.shape)  # and their grads in the grad arena
        self.precision = {'fp32': 0, 'bf16': 1, 'fp16': 2}[precision]
        self.desc = build_desc(model, self.layout, self.n_params, loss_type, beta, 0.999, 1e-8, self.precision)
        nbytes = lib().orn_engine_ws_bytes(byref(self.desc))
        if nbytes == 0:
            raise OrnError('orn_engine_ws_bytes: ' + _lib.last_error())
        self.ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self._h = c_void_p()
        check(lib().orn_engine_create(byref(self.desc), _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.adam_m),
                                      _lib.ptr(self.adam_v), _lib.ptr(self.ws), c_size_t(nbytes), byref(self._h)),
              'orn_engine_create')
        self.target_cache = bool(target_cache) and loss_type == 'Fusion6'
        self.tstats = None
        self.n_slots = n_slots
        self.stats_ring = torch.zeros(n_slots, 8, device=dev)
        self.cursor = torch.zeros(1, dtype=torch.int32, device=dev)
        self.sched = None
        self.frames = None
        self.embeds = None
        self.global_step = 0
        self.skipped_carry = 0                             # steps skipped by engines this one replaced (main_train fall-back)
        Hs, Ws = model.fc_h, model.fc_w
        for blk in model.layers:
            Hs, Ws = Hs * blk.stride, Ws * blk.stride
        self.out_hw = (Hs, Ws)
        # HIP graphs cannot be captured on the legacy default stream: the engine runs on its own
        # stream, ordered against the caller's current stream on entry and exit.
        self.stream = torch.cuda.Stream(device=dev)

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            try:
                lib().orn_engine_destroy(h)
                self._h = c_void_p()
            except Exception:          # interpreter shutdown: module globals may already be gone
                pass

    # ---- data ---------------------------------------------------------------------------------
    def set_video(self, frames: torch.Tensor, embeds: torch.Tensor):
        """frames [N,3,H,W] fp32 in [0,1] (resident in HBM), embeds [N,E] (PE of k/N)."""
        if tuple(frames.shape[1:]) != (3,) + self.out_hw:
            raise OrnError(f'frames {tuple(frames.shape)} do not match the decoder output {self.out_hw}')
        if embeds.shape != (frames.shape[0], self.desc.embed_len):
            raise OrnError(f'embeds {tuple(embeds.shape)} vs ({frames.shape[0]}, {self.desc.embed_len})')
        self.frames = frames.to(self.device, torch.float32).contiguous()
        self.embeds = embeds.to(self.device, torch.float32).contiguous()
        # Fusion6: the target side of the SSIM statistics, once per video (two valid-map planes per image plane: 2.9 GB for
        # 132 frames of 720p, 29 GB for 600 of 1080p, of 288 GB); the step then filters three maps instead of five.
        # Skipped when it does not fit beside the video with room to spare.
        self.tstats = None
        check(lib().orn_engine_set_target_stats(self._h, None), 'orn_engine_set_target_stats')
        if self.target_cache:
            n, _, H, W = self.frames.shape
            nbytes = lib().orn_loss_target_stats_bytes(n, 3, H, W)
            free, _ = torch.cuda.mem_get_info(self.device)
            if 0 < nbytes < free // 2:
                self.tstats = torch.empty(nbytes // 4, device=self.device)
                cur = torch.cuda.current_stream()
                check(lib().orn_loss_target_stats(_lib.ptr(self.frames), n, 3, H, W, _lib.ptr(self.tstats), c_void_p(cur.cuda_stream)),
                      'orn_loss_target_stats')
                check(lib().orn_engine_set_target_stats(self._h, _lib.ptr(self.tstats)), 'orn_engine_set_target_stats')

    def set_schedule(self, entries: Sequence[Tuple[int, int, float]]):
        """Upload the next run's per-step (frame, global step, lr) entries and rewind the cursor."""
        arr = make_schedule(entries)
        if self.frames is not None and len(entries) and (arr[:, 0].min() < 0 or arr[:, 0].max() >= self.frames.shape[0]):
            raise OrnError('schedule frame index out of range')
        if self.sched is None or self.sched.shape[0] < arr.shape[0]:
            self.sched = torch.zeros(max(arr.shape[0], 1), 4, dtype=torch.int32, device=self.device)
        self.sched[:arr.shape[0]].copy_(torch.from_numpy(arr), non_blocking=False)
        self.cursor.zero_()
        self._sched_len = arr.shape[0]

    # ---- stepping -----------------------------------------------------------------------------
    def run(self, n_steps: int, graph=None):
        """Enqueue `n_steps` optimiser steps consuming the uploaded schedule (no host sync).
        graph=None (default): orn_engine_train_steps -- plain stream launches, pipelined over the engine's second stream where the
        engine can (16-bit modes; include/orn.h); graph=True: hipGraph replay of the serial step; graph=False: one
        orn_engine_train_step call per step.  All three give bit-identical results."""
        if self.frames is None or self.sched is None:
            raise OrnError('set_video() and set_schedule() first')
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        st = c_void_p(self.stream.cuda_stream)
        if graph is None:
            check(lib().orn_engine_train_steps(self._h, _lib.ptr(self.frames), _lib.ptr(self.embeds), _lib.ptr(self.sched),
                                               _lib.ptr(self.cursor), _lib.ptr(self.stats_ring), c_int32(self.n_slots),
                                               c_int32(n_steps), st), 'orn_engine_train_steps')
        elif graph:
            check(lib().orn_engine_train_steps_graph(self._h, _lib.ptr(self.frames), _lib.ptr(self.embeds), _lib.ptr(self.sched),
                                                     _lib.ptr(self.cursor), _lib.ptr(self.stats_ring), c_int32(self.n_slots),
                                                     c_int32(n_steps), st), 'orn_engine_train_steps_graph')
        else:
            for _ in range(n_steps):
                check(lib().orn_engine_train_step(self._h, _lib.ptr(self.frames), _lib.ptr(self.embeds), _lib.ptr(self.sched),
                                                  _lib.ptr(self.cursor), _lib.ptr(self.stats_ring), c_int32(self.n_slots), st),
                      'orn_engine_train_step')
        cur.wait_stream(self.stream)
        self.global_step += n_steps

    def set_grad_mask(self, masks=None):
        """0/1 gradient masks per parameter name ({name: tensor like the parameter}; missing names = 1); None removes
        the mask.  Used by the prune fine-tune (main_eval.py:213-531)."""
        if masks is None:
            self._gmask = None
            check(lib().orn_engine_set_grad_mask(self._h, None), 'orn_engine_set_grad_mask')
            return
        gm = torch.ones(self.n_params, device=self.device)
        for k, m in masks.items():
            off, n = self.layout[k]
            gm[off:off + n] = m.to(self.device, torch.float32).reshape(-1)
        self._gmask = gm                                    # keep alive: the engine holds the raw pointer
        check(lib().orn_engine_set_grad_mask(self._h, _lib.ptr(gm)), 'orn_engine_set_grad_mask')

    def profile_step(self):
        """One eager optimiser step with HIP events around the conv launches, on the engine's stream.  Returns a dict of
        host lists / floats in ms: 'fwd'[i] forward conv of layer i, 'dgrad'[i] dgrad launch of layer i (fp32 mode: the
        layer's whole backward call), 'wgrad' the batched wgrad launch of the 16-bit layers, 'wgrad_reduce' its split-K
        reduction.  Consumes one schedule entry; synchronises."""
        if self.frames is None or self.sched is None:
            raise OrnError('set_video() and set_schedule() first')
        import ctypes
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        n = self.desc.n_layers
        ms = (ctypes.c_float * (2 * n + 2))()
        check(lib().orn_engine_profile_step(self._h, _lib.ptr(self.frames), _lib.ptr(self.embeds), _lib.ptr(self.sched),
                                            _lib.ptr(self.cursor), _lib.ptr(self.stats_ring), c_int32(self.n_slots), ms,
                                            c_void_p(self.stream.cuda_stream)), 'orn_engine_profile_step')
        cur.wait_stream(self.stream)
        self.global_step += 1
        v = [float(x) for x in ms]
        return {'fwd': v[:n], 'dgrad': v[n:2 * n], 'wgrad': v[2 * n], 'wgrad_reduce': v[2 * n + 1]}

    def applied_steps(self) -> int:
        """Optimiser steps that changed the parameters so far: enqueued steps minus the ones the non-finite guard skipped, in this
        engine (device counter) and in engines it replaced (`skipped_carry`, set by main_train's precision fall-back).  This is
        torch.optim.Adam's 'step' of a checkpoint (include/orn.h).  Synchronises."""
        return int(self.global_step - self.skipped_carry - self.scale_state()['skipped'])

    def scale_state(self) -> dict:
        """Dynamic loss scale / non-finite guard of the engine (device state; synchronises): scale, ceiling, flag, steps
        skipped so far, clean steps since the last change, halvings."""
        out = (ctypes.c_float * 8)()
        check(lib().orn_engine_scale_state(self._h, out), 'orn_engine_scale_state')
        return {'scale': float(out[0]), 'ceiling': float(out[2]), 'flag': int(out[3]), 'skipped': int(out[4]),
                'good': int(out[5]), 'backoffs': int(out[6])}

    def set_grad_scale(self, scale: float, ceiling: float = 0.0):
        """Override the live gradient scale of the 16-bit modes (and its ceiling if > 0)."""
        check(lib().orn_engine_set_grad_scale(self._h, ctypes.c_float(scale), ctypes.c_float(ceiling)), 'orn_engine_set_grad_scale')

    def stats(self, n: int) -> torch.Tensor:
        """[n,8] host tensor of the last run's first n steps: loss, L1, MSE, SSIM, PSNR, lr, frame, step."""
        return self.stats_ring[:n].cpu()

    def decode(self, embed: torch.Tensor) -> torch.Tensor:
        """Forward only: embed [E] or [1,E] -> image [1,3,H,W]."""
        embed = embed.to(self.device, torch.float32).contiguous().view(-1)
        img = torch.empty(1, 3, *self.out_hw, device=self.device)
        check(lib().orn_engine_decode(self._h, _lib.ptr(embed), _lib.ptr(img), _lib.stream()), 'orn_engine_decode')
        return img

    def engine_fused_kernel(self, layer: int):
        """(Wf, bf) of block `layer` exactly as the ENGINE's last merge left them in its workspace (copies; ERB only)."""
        wf, bf = c_void_p(), c_void_p()
        check(lib().orn_engine_fused_kernel(self._h, layer, byref(wf), byref(bf)), 'orn_engine_fused_kernel')
        shape = tuple(dict(self.model.named_parameters())[f'layers.{layer}.rbr_3x3_branch.weight'].shape)
        n = shape[0] * shape[1] * 9
        torch.cuda.synchronize()
        base = self.ws.data_ptr()

        def view(ptr, cnt):                       # the pointers lie inside the engine's workspace tensor
            off = ptr - base
            assert 0 <= off and off + 4 * cnt <= self.ws.numel(), 'fused kernel pointer outside the workspace'
            return self.ws[off:off + 4 * cnt].view(torch.float32).clone()
        return view(wf.value, n).view(shape), view(bf.value, shape[0])

    def fused_kernel(self, layer: int):
        """(Wf, bf) of block `layer` as produced by the last merge (model.py:450-478), as tensors."""
        blk = self.model.layers[layer]
        if self.desc.erb:
            with torch.no_grad():
                return blk.get_equivalent_kernel_bias()
        conv = blk.rbr_reparam if blk.deploy else blk.branch
        return conv.weight, conv.bias
