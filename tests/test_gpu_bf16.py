"""GPU tests of the bf16 MFMA fast path (C-ABI hooks orn_conv3x3_ps_silu_{fwd,bwd}_bf16) against the
fp32 oracle.  Tolerances are bf16 tolerances: inputs/weights rounded to 8 significant bits, fp32
accumulation; stated per assertion."""
import math
import os
from ctypes import c_size_t

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def orn():
    import orn_amd
    from orn_amd import ops, model, utils, engine  # noqa: F401
    orn_amd._lib.lib()
    return orn_amd


def _half_round(t, half):
    return t.to(torch.bfloat16 if half == 'bf16' else torch.float16).to(torch.float32)


@pytest.mark.parametrize('half', ['bf16', 'fp16'])
@pytest.mark.parametrize('C,O,H,W,s', [(96, 384, 8, 32, 2), (96, 384, 36, 64, 2), (96, 384, 45, 80, 2), (96, 384, 13, 37, 2),
                                       (96, 1152, 9, 33, 3), (96, 864, 10, 33, 3),
                                       # >= 128 pixel tiles: the forward takes the two-work-groups-per-CU form (whole and ragged tiles, s = 2 / 3)
                                       (96, 384, 64, 512, 2), (96, 384, 67, 500, 2), (96, 864, 66, 480, 3),
                                       # >= 400 tiles, ragged in both directions: the same kernel for blocks that also write the activation copy
                                       (96, 384, 100, 1030, 2)])
def test_bf16_block_fwd_bwd(orn, C, O, H, W, s, half):
    """conv3x3+PixelShuffle+SiLU fwd, and dbias / wgrad / dgrad, on 16-bit MFMA -- the bf16 build AND the IEEE-half build
    the engine's fp16 mode (bench.py's headline) launches -- vs the CPU oracle run on the SAME 16-bit-rounded inputs
    (isolates kernel correctness from input quantisation): exact integer-free check would hide layout bugs, so data are
    asymmetric random.  Tolerances: outputs are stored in 16 bit (2^-9 bf16 / 2^-12 fp16 relative) on fp32 accumulation."""
    from oracle import cpu_ref
    L, P, st = orn._lib.lib(), orn._lib.ptr, orn._lib.stream
    _bf16_round = lambda t: _half_round(t, half)
    fwd_fn = L.orn_conv3x3_ps_silu_fwd_bf16 if half == 'bf16' else L.orn_conv3x3_ps_silu_fwd_f16
    bwd_fn = L.orn_conv3x3_ps_silu_bwd_bf16 if half == 'bf16' else L.orn_conv3x3_ps_silu_bwd_f16
    tol = 5e-3 if half == 'bf16' else 1e-3
    gen = torch.Generator().manual_seed(C + O + H * W)
    x = _bf16_round(torch.randn(1, C, H, W, generator=gen))
    wf = _bf16_round(torch.randn(O, C, 3, 3, generator=gen) / math.sqrt(9 * C))
    bf = torch.randn(O, generator=gen) * 0.1
    Cn = O // (s * s)
    da = torch.randn(1, Cn, H * s, W * s, generator=gen)
    rx, rw, rb = (t.clone().requires_grad_(True) for t in (x, wf, bf))
    torch.set_num_threads(8)
    y = torch.nn.functional.conv2d(rx, rw, rb, padding=1)
    zr = torch.nn.functional.pixel_shuffle(y, s)
    ar = torch.nn.functional.silu(zr)
    nb = L.orn_conv3x3_ps_silu_bf16_ws_bytes(C, O, H, W, s)
    ws = torch.zeros(nb, dtype=torch.uint8, device='cuda')
    xd, wd, bd = x.cuda(), wf.cuda(), bf.cuda()
    z = torch.empty(1, Cn, H * s, W * s, device='cuda')
    a = torch.empty_like(z)
    orn._lib.check(fwd_fn(P(xd), P(wd), P(bd), C, O, H, W, s, P(z), P(a), P(ws), c_size_t(nb), st()))
    np.testing.assert_allclose(z.cpu().numpy(), zr.detach().numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(a.cpu().numpy(), ar.detach().numpy(), rtol=tol, atol=tol)
    # the last block's form (no activation copy): on >= 128 pixel tiles this is the two-work-groups-per-CU kernel
    z2 = torch.full_like(z, float('nan'))
    orn._lib.check(fwd_fn(P(xd), P(wd), P(bd), C, O, H, W, s, P(z2), None, P(ws), c_size_t(nb), st()))
    np.testing.assert_allclose(z2.cpu().numpy(), zr.detach().numpy(), rtol=tol, atol=tol)
    if C != 96:
        return
    # backward with the oracle's z and the same bf16-rounded dy the kernel sees
    zq = _bf16_round(zr.detach())
    dz = _bf16_round(da * (torch.sigmoid(zq) * (1 + zq * (1 - torch.sigmoid(zq)))))
    dy = torch.nn.functional.pixel_unshuffle(dz, s)
    gx, gw = torch.autograd.grad(y, (rx, rw), dy)
    gb = dy.sum(dim=(0, 2, 3))
    dx = torch.empty(1, C, H, W, device='cuda')
    dwf = torch.empty(O, C, 3, 3, device='cuda')
    dbf = torch.empty(O, device='cuda')
    zd, dad = zr.detach().cuda().contiguous(), da.cuda()       # keep alive: ptr() does not hold a reference
    orn._lib.check(bwd_fn(P(xd), P(wd), P(zd), P(dad), C, O, H, W, s,
                                                  P(dx), P(dwf), P(dbf), P(ws), c_size_t(nb), st()))
    torch.cuda.synchronize()
    sc = float(gw.abs().max())
    np.testing.assert_allclose(dwf.cpu().numpy(), gw.numpy(), rtol=2e-3, atol=2e-3 * sc)
    np.testing.assert_allclose(dbf.cpu().numpy(), gb.numpy(), rtol=2e-3, atol=2e-3 * float(gb.abs().max()))
    np.testing.assert_allclose(dx.cpu().numpy(), gx.numpy(), rtol=2e-3, atol=2e-3 * float(gx.abs().max()))


def _mk(orn, fc, strides, lower_width, bt='ERB'):
    torch.manual_seed(1)
    return orn.model.Generator(embed_length=80, stem_dim_num='32_1', fc_hw_dim=fc, expansion=1, num_blocks=1, norm='none',
                               act='swish', bias=True, reduction=2, conv_type='conv', stride_list=strides, sin_res=True,
                               lower_width=lower_width, sigmoid=False, deploy=False, branch_type=bt)


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize('half', ['bf16', 'fp16'])
@pytest.mark.parametrize('fc,strides,lw,bt', [('3_4_96', [2, 2], 96, 'ERB'), ('2_3_26', [5, 2, 2], 96, 'ERB'), ('2_3_48', [5, 3, 2], 96, 'ERB'),
                                               ('3_4_96', [2, 2, 2], 96, 'NeRV_vanilla')])
def test_bf16_engine_vs_fp32_engine(orn, fc, strides, lw, bt, half):
    """One optimiser step of the 16-bit engine (bf16 or IEEE half) vs the fp32 engine (itself pinned to the
    oracle): same loss to 2e-3 relative, every gradient tensor within 3 % (bf16) / 0.6 % (fp16) relative L2
    (16-bit activations / weights, fp32 accumulation), decode output within 2e-2 / 4e-3 abs."""
    from oracle import cpu_ref
    res = {}
    n_frames = 3
    hw = None
    for prec in ('fp32', half):
        gen = _mk(orn, fc, strides, lw, bt)
        eng = orn.engine.TrainEngine(gen, loss_type='Fusion6', beta=0.5, precision=prec)
        if hw is None:
            hw = eng.out_hw
            frames = cpu_ref.synthetic_video(n_frames, hw[0], hw[1], seed=5)
            embeds = cpu_ref.positional_encoding(torch.tensor([k / n_frames for k in range(n_frames)]), 1.25, 40)
        eng.set_video(frames, embeds)
        eng.set_schedule([(1, 1, 0.0)])          # lr 0: parameters stay put, gradients are what we compare
        eng.run(1, graph=(prec != 'fp32'))
        torch.cuda.synchronize()
        grads = {k: eng.grads[off:off + n].clone().cpu() for k, (off, n) in eng.layout.items()}
        img = eng.decode(embeds[2]).cpu()
        res[prec] = (eng.stats(1)[0].clone(), grads, img)
    sf, gf, imf = res['fp32']
    sb, gb, imb = res[half]
    tight = (half == 'fp16')
    assert abs(sb[0] - sf[0]) <= (3e-4 if tight else 2e-3) * abs(sf[0]), (sb, sf)
    assert abs(sb[4] - sf[4]) <= (0.01 if tight else 0.05), (sb, sf)          # PSNR of the prediction, dB
    assert torch.max(torch.abs(imb - imf)) < (4e-3 if tight else 2e-2)
    worst = max((_rel(gb[k], gf[k]), k) for k in gf if gf[k].norm() > 0)
    assert worst[0] < (6e-3 if tight else 3e-2), worst


@pytest.mark.parametrize('half,cfg', [('bf16', '720p'), ('fp16', '720p'), ('fp16', '1080p')])
def test_bf16_engine_720p_decode_and_step(orn, half, cfg):
    """BASELINE config 2 (720p, 9_16_26, strides 5 2 2 2 2) and config 3 (1080p, 9_16_48, strides 5 3 2 2 2: stride-3
    block, 135x240 / 270x480 / 540x960 conv inputs that are not multiples of the pixel tile) on the 16-bit path: decode
    agrees with the fp32 path (PSNR between the two decoders > 50 / 65 dB) and a training step runs with finite loss
    and gradients close to fp32."""
    import bench
    outs = {}
    kw = dict(frames=8) if cfg == '1080p' else {}
    for prec in ('fp32', half):
        eng = bench.make_engine(seed=1234, precision=prec, cfg=bench.CONFIGS[cfg], **kw)
        eng.set_schedule([(7, 1, 0.0)])
        eng.run(1, graph=True)
        torch.cuda.synchronize()
        outs[prec] = (eng.decode(eng.embeds[7]).cpu(), eng.stats(1)[0].clone(),
                      {k: eng.grads[off:off + n].clone().cpu() for k, (off, n) in eng.layout.items()})
        del eng
        torch.cuda.empty_cache()
    imf, sf, gf = outs['fp32']
    imb, sb, gb = outs[half]
    mse = float(((imf - imb) ** 2).mean())
    assert -10 * math.log10(mse) > (65.0 if half == 'fp16' else 50.0), mse
    assert torch.isfinite(sb).all() and abs(sb[0] - sf[0]) <= 2e-3 * abs(sf[0])
    worst = max((_rel(gb[k], gf[k]), k) for k in gf if gf[k].norm() > 0)
    assert worst[0] < (1e-2 if half == 'fp16' else 5e-2), worst


@pytest.mark.parametrize('switch', ['ORN_HEAD_FUSED', 'ORN_FWD2_APAD', 'ORN_FWD_FORM1', 'ORN_DGRAD_FORM1'])
def test_kernel_selection_switches_keep_the_720p_step_correct(switch):
    """The launchers pick between kernel forms by shape; the forms that are NOT the default at 720p stay selectable through
    environment switches read once per process (tools/probes A/B runs): the head inside the last block's epilogue, the
    two-work-groups-per-CU forward for blocks that also write the activation copy, and the first forms of forward and dgrad.
    Each must pass the 720p engine-vs-fp32 check of this file in a process of its own."""
    import subprocess
    import sys
    env = dict(os.environ, **{switch: '1'})
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-m', 'gpu', '-x', '-p', 'no:cacheprovider',
                        '-k', 'test_bf16_engine_720p_decode_and_step and fp16-720p'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and '1 passed' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize('prec', ['fp32', 'bf16', 'fp16'])
def test_engine_is_run_to_run_deterministic(orn, prec):
    """No atomics, fixed-order reductions: two engines from the same seed must agree bit for bit after
    several optimiser steps (parameters, Adam state and the per-step loss ring)."""
    from oracle import cpu_ref
    outs = []
    for rep in range(2):
        gen = _mk(orn, '2_3_26', [5, 2, 2], 96, 'ERB')
        eng = orn.engine.TrainEngine(gen, loss_type='Fusion6', beta=0.5, precision=prec)
        hw = eng.out_hw
        frames = cpu_ref.synthetic_video(4, hw[0], hw[1], seed=5)
        embeds = cpu_ref.positional_encoding(torch.tensor([k / 4 for k in range(4)]), 1.25, 40)
        eng.set_video(frames, embeds)
        eng.set_schedule([(k % 4, k + 1, 5e-4) for k in range(12)])
        eng.run(12, graph=True)
        torch.cuda.synchronize()
        outs.append((eng.params.clone(), eng.adam_v.clone(), eng.stats(12).clone()))
        # poison freed memory so that a read of uninitialised scratch would differ between the two runs
        del eng, gen
        junk = torch.full((64 * 1024 * 1024,), float(rep + 1), device='cuda')
        del junk
    rows = [i for i in range(12) if not torch.equal(outs[0][2][i], outs[1][2][i])]
    assert not rows, ('loss ring rows differ', rows, outs[0][2][rows[0]].tolist(), outs[1][2][rows[0]].tolist())
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])


def test_profile_step_and_grad_mask(orn):
    """orn_engine_profile_step returns a positive duration for every layer's forward conv launch and advances training
    like a normal step; orn_engine_set_grad_mask freezes exactly the masked entries (Adam sees a zero gradient)."""
    from oracle import cpu_ref
    gen = _mk(orn, '2_3_26', [5, 2, 2], 96, 'ERB')
    eng = orn.engine.TrainEngine(gen, loss_type='Fusion6', beta=0.5, precision='fp16')
    hw = eng.out_hw
    eng.set_video(cpu_ref.synthetic_video(3, hw[0], hw[1], seed=5),
                  cpu_ref.positional_encoding(torch.tensor([k / 3 for k in range(3)]), 1.25, 40))
    eng.set_schedule([(k % 3, k + 1, 5e-4) for k in range(4)])
    name = 'layers.1.rbr_3x3_branch.weight'
    off, n = eng.layout[name]
    mask = (torch.rand(n, generator=torch.Generator().manual_seed(0)) > 0.5).float()
    before = eng.params[off:off + n].clone()
    eng.set_grad_mask({name: mask.view(tuple(dict(gen.named_parameters())[name].shape))})      # fresh Adam state: m = v = 0
    eng.run(2, graph=True)
    torch.cuda.synchronize()
    after = eng.params[off:off + n]
    frozen = mask.cuda() == 0
    assert torch.equal(after[frozen], before[frozen]) and not torch.equal(after[~frozen], before[~frozen])
    eng.set_grad_mask(None)
    ms = eng.profile_step()
    assert len(ms['fwd']) == 3 and all(0.0 < m < 5.0 for m in ms['fwd']), ms
    assert all(0.0 < m < 5.0 for m in ms['dgrad']) and 0.0 < ms['wgrad'] < 5.0 and 0.0 < ms['wgrad_reduce'] < 5.0, ms
    eng.run(1, graph=True)
    torch.cuda.synchronize()
    assert not torch.equal(eng.params[off:off + n][frozen], before[frozen])
    assert int(eng.stats(4)[3, 7]) == 4
