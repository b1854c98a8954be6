"""GPU parity tests: every hot-path op called through the C ABI of liborn.so (via the ctypes/autograd
glue) against the CPU oracle and the golden vectors captured from the reference.  Run on the GPU box
with `pytest -m gpu`."""
import math

import numpy as np
import pytest
import torch

from helpers import ERB_KEYS, erb_inputs

pytestmark = pytest.mark.gpu

T = torch.from_numpy


@pytest.fixture(scope='module')
def orn():
    import orn_amd
    from orn_amd import ops, model, utils, engine  # noqa: F401
    orn_amd._lib.lib()
    return orn_amd


def cu(x):
    if isinstance(x, np.ndarray):
        x = T(x)
    return x.to('cuda')


# ---------------------------------------------------------------- A1 --------------------------
def test_pe_matches_reference(orn, golden):
    g = golden('utils')
    out = orn.ops.pe_forward(cu(g['pe/pos']), 1.25, 40).cpu().numpy()
    assert out.shape == (132, 80)
    # fp32 argument identical; device sinf/cosf vs the reference's CPU libm: <= 2 ulp at |x|<=1
    assert np.max(np.abs(out - g['pe/batched'])) <= 1.3e-7
    pe = orn.utils.PositionalEncoding('1.25_40')
    assert pe.embed_length == 80
    assert np.array_equal(pe(T(g['pe/pos'][1:2])).cpu().numpy(), out[1:2])


# ---------------------------------------------------------------- A2 --------------------------
@pytest.mark.parametrize('B,E,Hd,Nout', [(1, 80, 512, 3744), (2, 80, 32, 96)])
def test_stem_fwd_bwd(orn, B, E, Hd, Nout):
    from oracle import cpu_ref
    gen = torch.Generator().manual_seed(10)
    e = torch.randn(B, E, generator=gen)
    w0 = torch.randn(Hd, E, generator=gen) / math.sqrt(E)
    b0 = torch.randn(Hd, generator=gen) * 0.1
    w1 = torch.randn(Nout, Hd, generator=gen) / math.sqrt(Hd)
    b1 = torch.randn(Nout, generator=gen) * 0.1
    dh2 = torch.randn(B, Nout, generator=gen)
    ref_in = [t.clone().requires_grad_(True) for t in (w0, b0, w1, b1)]
    ref = cpu_ref.stem_forward(e, *ref_in)
    (ref * dh2).sum().backward()
    dev_in = [cu(t).requires_grad_(True) for t in (w0, b0, w1, b1)]
    out = orn.ops.StemFn.apply(cu(e), *dev_in)
    (out * cu(dh2)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=2e-6)
    for a, b, name in zip(dev_in, ref_in, ('w0', 'b0', 'w1', 'b1')):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=3e-5, atol=3e-6, err_msg=name)


# ---------------------------------------------------------------- A3 --------------------------
@pytest.mark.parametrize('C,O,seed', [(6, 16, 11), (26, 52, 12), (26, 650, 21), (26, 384, 22), (96, 384, 23),
                                      (48, 1200, 24), (48, 864, 25)])
def test_merge_fwd_bit_exact(orn, golden, C, O, seed):
    """HIP merge == oracle/merge_ref.c bit for bit (specified fmaf order); <= 1.5e-7 of the reference."""
    from oracle import c_oracle
    w = erb_inputs(C, O, seed)
    wf_c, bf_c, T_c = c_oracle.merge_fwd(*[w[k].numpy() for k in ERB_KEYS])
    wf, bf = orn.ops.ErbMergeFn.apply(*[cu(w[k]) for k in ERB_KEYS])
    assert np.array_equal(wf.cpu().numpy(), wf_c)
    assert np.array_equal(bf.cpu().numpy(), bf_c)
    g = golden('merge')
    if C <= 26 and O <= 52:
        tag = f'C{C}_O{O}'
        assert np.max(np.abs(wf.cpu().numpy() - g[f'{tag}/Wf'])) <= 1e-7
        assert np.array_equal(bf.cpu().numpy(), g[f'{tag}/bf'])
    else:
        tag = f'real_C{C}_O{O}'
        assert np.max(np.abs(wf.cpu().numpy().reshape(-1)[g[f'{tag}/idx']] - g[f'{tag}/Wf_samples'])) <= 1.5e-7
        assert np.array_equal(bf.cpu().numpy(), g[f'{tag}/bf'])


@pytest.mark.parametrize('C,O,seed', [(7, 25, 31), (5, 99, 32), (33, 65, 33), (3, 130, 34)])
def test_merge_fwd_bit_exact_odd_shapes(orn, C, O, seed):
    """Rows of odd length: the merge GEMM's 16-byte buffer loads start at every 4-byte alignment, the K tail (K % 32 != 0,
    K % 4 != 0) is fed by reads beyond the end of B (zeros) and ragged tiles read beyond M / N.  Bit-equal to oracle/merge_ref.c."""
    from oracle import c_oracle
    w = erb_inputs(C, O, seed)
    wf_c, bf_c, T_c = c_oracle.merge_fwd(*[w[k].numpy() for k in ERB_KEYS])
    wf, bf = orn.ops.ErbMergeFn.apply(*[cu(w[k]) for k in ERB_KEYS])
    assert np.array_equal(wf.cpu().numpy(), wf_c)
    assert np.array_equal(bf.cpu().numpy(), bf_c)


@pytest.mark.parametrize('tag,C,O', [('C6_O16', 6, 16), ('C26_O52', 26, 52)])
def test_merge_bwd_golden(orn, golden, tag, C, O):
    g = golden('merge')
    w = {k: cu(g[f'{tag}/in/{k}']).requires_grad_(True) for k in ERB_KEYS}
    wf, bf = orn.ops.ErbMergeFn.apply(*[w[k] for k in ERB_KEYS])
    ((wf * cu(g[f'{tag}/G'])).sum() + (bf * cu(g[f'{tag}/dbf'])).sum()).backward()
    for k in ERB_KEYS:
        np.testing.assert_allclose(w[k].grad.cpu().numpy(), g[f'{tag}/grad/{k}'], rtol=2e-5, atol=2e-6, err_msg=k)


@pytest.mark.parametrize('C,O,seed', [(26, 650, 21), (96, 384, 23)])
def test_merge_bwd_real_shapes_vs_oracle(orn, C, O, seed):
    from oracle import cpu_ref
    w = erb_inputs(C, O, seed)
    gen = torch.Generator().manual_seed(seed + 1)
    G = torch.randn(O, C, 3, 3, generator=gen)
    dbf = torch.randn(O, generator=gen)
    ref = cpu_ref.erb_merge_backward_closed_form(G.double(), dbf.double(),
                                                 w['rbr_1x1_3x3_1x1_branch_1x1_1.weight'].double(),
                                                 w['rbr_1x1_3x3_1x1_branch_3x3.weight'].double(),
                                                 w['rbr_1x1_3x3_1x1_branch_1x1_2.weight'].double())
    wd = {k: cu(w[k]).requires_grad_(True) for k in ERB_KEYS}
    wf, bf = orn.ops.ErbMergeFn.apply(*[wd[k] for k in ERB_KEYS])
    ((wf * cu(G)).sum() + (bf * cu(dbf)).sum()).backward()
    for k in ERB_KEYS:
        r = ref[k].float().numpy()
        np.testing.assert_allclose(wd[k].grad.cpu().numpy(), r, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(r).max()), err_msg=k)


# ---------------------------------------------------------------- A4 --------------------------
@pytest.mark.parametrize('s', [2, 3, 5])
def test_block_golden(orn, golden, s):
    """NeRVBlock fwd/bwd (ERB online merge -> conv -> PixelShuffle -> SiLU) vs the reference."""
    g = golden('block')
    tag = f's{s}'
    blk = orn.model.NeRVBlock(ngf=6, new_ngf=4, stride=s, bias=True, norm='none', act='swish', deploy=False,
                              conv_type='conv', branch_type='ERB')
    blk.load_state_dict({k: T(g[f'{tag}/erb/in/{k}']) for k in ERB_KEYS})
    blk = blk.cuda()
    x = cu(g[f'{tag}/x']).requires_grad_(True)
    a = blk(x)
    np.testing.assert_allclose(a.detach().cpu().numpy(), g[f'{tag}/erb/a'], rtol=1e-5, atol=2e-6)
    (a * cu(g[f'{tag}/da'])).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[f'{tag}/erb/dx'], rtol=2e-5, atol=5e-6)
    params = dict(blk.named_parameters())
    for k in ERB_KEYS:
        np.testing.assert_allclose(params[k].grad.cpu().numpy(), g[f'{tag}/erb/grad/{k}'], rtol=5e-5, atol=1e-5, err_msg=k)
    # deploy: same forward, reference key layout (model.py:395-448)
    blk.switch_to_deploy()
    assert sorted(blk.state_dict().keys()) == ['rbr_reparam.bias', 'rbr_reparam.weight']
    assert np.max(np.abs(blk.rbr_reparam.weight.detach().cpu().numpy() - g[f'{tag}/deploy/weight'])) <= 1e-7
    with torch.no_grad():
        a2 = blk(x.detach())
    assert torch.equal(a2, a.detach())          # deploy forward == train forward, bit-identical
    blk.switch_to_deploy()                      # idempotent


@pytest.mark.parametrize('B,C,O,H,W,s', [(1, 26, 104, 45, 80, 2), (1, 96, 384, 36, 64, 2), (2, 26, 650, 9, 16, 5),
                                         (1, 48, 864, 23, 41, 3), (1, 7, 36, 1, 1, 3)])
def test_conv_ps_silu_vs_oracle(orn, B, C, O, H, W, s):
    """Real channel counts, ragged H/W (not multiples of the 8x32 tile), batch 2, 1x1 image."""
    from oracle import cpu_ref
    gen = torch.Generator().manual_seed(C * 1000 + O)
    x = torch.randn(B, C, H, W, generator=gen)
    wf = torch.randn(O, C, 3, 3, generator=gen) / math.sqrt(9 * C)
    bf = torch.randn(O, generator=gen) * 0.1
    da = torch.randn(B, O // (s * s), H * s, W * s, generator=gen)
    rx, rw, rb = (t.clone().requires_grad_(True) for t in (x, wf, bf))
    torch.set_num_threads(8)
    ref = cpu_ref.block_forward(rx, rw, rb, s)
    (ref * da).sum().backward()
    dx, dw, db = (cu(t).requires_grad_(True) for t in (x, wf, bf))
    out = orn.ops.ConvPsSiluFn.apply(dx, dw, db, s)
    (out * cu(da)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(dx.grad.cpu().numpy(), rx.grad.numpy(), rtol=1e-4, atol=1e-4)
    scale = float(rw.grad.abs().max())
    np.testing.assert_allclose(dw.grad.cpu().numpy(), rw.grad.numpy(), rtol=1e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(db.grad.cpu().numpy(), rb.grad.numpy(), rtol=1e-4, atol=2e-5 * float(rb.grad.abs().max()))


def test_conv_linearity_full_size(orn):
    """Size-independent property at BASELINE's full L4 geometry (C=96,O=384,360x640,s=2):
    conv(x1 + x2) - bias == conv(x1) + conv(x2) - 2 bias (pre-activation, checked via z)."""
    from ctypes import c_int
    lib, ptr, stream = orn._lib.lib(), orn._lib.ptr, orn._lib.stream
    gen = torch.Generator(device='cuda').manual_seed(3)
    C, O, H, W, s = 96, 384, 360, 640, 2
    x1 = torch.randn(1, C, H, W, device='cuda', generator=gen)
    x2 = torch.randn(1, C, H, W, device='cuda', generator=gen)
    wf = torch.randn(O, C, 3, 3, device='cuda', generator=gen) / math.sqrt(9 * C)
    bf = torch.zeros(O, device='cuda')
    zs = []
    for x in (x1, x2, x1 + x2):
        z = torch.empty(1, O // 4, H * 2, W * 2, device='cuda')
        a = torch.empty_like(z)
        orn._lib.check(lib.orn_conv3x3_ps_silu_fwd(ptr(x), ptr(wf), ptr(bf), 1, C, O, H, W, s, ptr(z), ptr(a), stream()))
        zs.append(z)
        assert torch.allclose(a, torch.nn.functional.silu(z), rtol=1e-5, atol=1e-6)
    err = (zs[2] - (zs[0] + zs[1])).abs().max().item()
    assert err < 2e-5, err
    # spot-check 64 random output pixels against a direct fp64 evaluation
    idx = torch.randint(0, H * W, (64,), generator=torch.Generator().manual_seed(1))
    xp = torch.nn.functional.pad(x1[0].double().cpu(), (1, 1, 1, 1))
    wd = wf.double().cpu()
    z0 = zs[0].cpu()
    for p in idx.tolist():
        h, w_ = divmod(p, W)
        patch = xp[:, h:h + 3, w_:w_ + 3]
        y = (wd * patch[None]).sum(dim=(1, 2, 3))        # [O]
        got = torch.stack([z0[0, o // 4, h * 2 + (o % 4) // 2, w_ * 2 + (o % 2)] for o in range(O)])
        assert torch.max(torch.abs(got.double() - y)) < 5e-5


# ---------------------------------------------------------------- A5 --------------------------
@pytest.mark.parametrize('B,C,H,W,sig', [(1, 96, 24, 40, False), (2, 8, 12, 16, False), (1, 16, 7, 9, True)])
def test_head_fwd_bwd(orn, B, C, H, W, sig):
    from oracle import cpu_ref
    gen = torch.Generator().manual_seed(50 + C)
    a = torch.randn(B, C, H, W, generator=gen)
    w = torch.randn(3, C, 1, 1, generator=gen) / math.sqrt(C)
    b = torch.randn(3, generator=gen) * 0.1
    do = torch.randn(B, 3, H, W, generator=gen)
    ra, rw, rb = (t.clone().requires_grad_(True) for t in (a, w, b))
    ref = cpu_ref.head_forward(ra, rw, rb, sig)
    (ref * do).sum().backward()
    da, dw, db = (cu(t).requires_grad_(True) for t in (a, w, b))
    out = orn.ops.HeadFn.apply(da, dw, db, sig)
    (out * cu(do)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(da.grad.cpu().numpy(), ra.grad.numpy(), rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(dw.grad.cpu().numpy(), rw.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(db.grad.cpu().numpy(), rb.grad.numpy(), rtol=2e-4, atol=2e-5)


# ---------------------------------------------------------------- A7 / A10 --------------------
@pytest.mark.parametrize('lt', ['L2', 'L1', 'Fusion6'])
@pytest.mark.parametrize('B,H,W', [(1, 45, 80), (2, 30, 37), (1, 90, 160)])
def test_loss_fwd_bwd(orn, lt, B, H, W):
    """Loss value, PSNR and dL/dpred vs the oracle (SSIM term: definition-level parity, see oracle)."""
    from oracle import cpu_ref
    gen = torch.Generator().manual_seed(H * W + B)
    t = torch.rand(B, 3, H, W, generator=gen)
    p = (t + 0.1 * torch.randn(B, 3, H, W, generator=gen)).clamp(0, 1)
    rp = p.clone().double().requires_grad_(True)
    rl = cpu_ref.loss_fn(rp, t.double(), lt)
    rl.backward()
    dp = cu(p).requires_grad_(True)
    class A: loss_type = lt
    loss = orn.utils.loss_fn(dp, cu(t), A)
    loss.backward()
    assert abs(loss.item() - rl.item()) <= 2e-6 + 1e-5 * abs(rl.item())
    g_ref = rp.grad.float().numpy()
    np.testing.assert_allclose(dp.grad.cpu().numpy(), g_ref, rtol=2e-3, atol=2e-4 * np.abs(g_ref).max())
    ps = orn.utils.psnr_fn([dp.detach()], [cu(t)]).cpu().numpy()
    np.testing.assert_allclose(ps, cpu_ref.psnr_fn([p], [t]).numpy(), rtol=1e-5)
    assert ps.shape == (B, 1)


def test_loss_golden_plain(orn, golden):
    g = golden('utils')
    a, b = cu(g['psnr/0/a']), cu(g['psnr/0/b'])
    for lt in ('L2', 'L1'):
        st, _ = orn.ops.loss_stats(a, b, lt, want_grad=False)
        assert abs(st[0].item() - float(g[f'loss/{lt}'])) <= 1e-7
    for i in range(3):
        ps = orn.utils.psnr_fn([cu(g[f'psnr/{i}/a'])], [cu(g[f'psnr/{i}/b'])]).cpu().numpy()
        np.testing.assert_allclose(ps, g[f'psnr/{i}/out'], rtol=2e-6)


# ---------------------------------------------------------------- N3 --------------------------
@pytest.mark.parametrize('B,H,W', [(1, 180, 320), (2, 161, 177), (1, 720, 1280)])
def test_msssim(orn, B, H, W):
    """msssim_fn (utils.py:201-211) vs the oracle's restatement of pytorch_msssim.ms_ssim (parity unpinned: the
    package is not in the reference tree).  fp32 tolerance 2e-5 absolute on a value in [0, 1]."""
    from oracle import cpu_ref
    gen = torch.Generator().manual_seed(H + W)
    t = torch.rand(B, 3, H, W, generator=gen)
    t = torch.nn.functional.avg_pool2d(t, 5, 1, 2)                     # some structure so cs is not ~0
    p = (t + 0.05 * torch.randn(B, 3, H, W, generator=gen)).clamp(0, 1)
    ref = cpu_ref.ms_ssim(p.double(), t.double()).item()
    out = orn.utils.msssim_fn([cu(p)], [cu(t)])
    assert out.shape == (B, 1)
    assert abs(out[0, 0].item() - ref) <= 2e-5, (out[0, 0].item(), ref)
    small = orn.utils.msssim_fn([cu(p[..., :100, :100])], [cu(t[..., :100, :100])])
    assert float(small.abs().max()) == 0.0                              # H < 160 -> 0, as the reference


# ---------------------------------------------------------------- A9 --------------------------
def test_adam(orn):
    from oracle import cpu_ref
    gen = torch.Generator().manual_seed(6)
    n = 1000 * 4 + 3
    p0 = torch.randn(n, generator=gen)
    p, m, v = p0.clone(), torch.zeros(n), torch.zeros(n)
    dp, dm, dv = cu(p0.clone()), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    for step in range(1, 6):
        gr = torch.randn(n, generator=gen) * 10 ** (-step)
        cpu_ref.adam_step(p, gr, m, v, step, 5e-4 * step, 0.5)
        orn.ops.adam_step_(dp, cu(gr), dm, dv, 5e-4 * step, step, 0.5)
        np.testing.assert_allclose(dp.cpu().numpy(), p.numpy(), rtol=3e-6, atol=1e-7)
        np.testing.assert_allclose(dv.cpu().numpy(), v.numpy(), rtol=3e-6, atol=1e-30)


# ---------------------------------------------------------------- A6 + whole model ------------
@pytest.mark.parametrize('bt', ['ERB', 'NeRV_vanilla'])
def test_tiny_generator_golden(orn, golden, bt):
    """Generator(**kargs) mirror: seeded init, state-dict keys, forward and all grads vs the reference."""
    g = golden('generator')
    tag = f'tiny_{bt}'
    torch.manual_seed(1)
    gen = orn.model.Generator(embed_length=80, stem_dim_num='32_1', fc_hw_dim='3_4_8', expansion=1, num_blocks=1,
                              norm='none', act='swish', bias=True, reduction=2, conv_type='conv', stride_list=[2, 2],
                              sin_res=True, lower_width=8, sigmoid=False, deploy=False, branch_type=bt)
    sd = gen.state_dict()
    assert list(sd.keys()) == list(g[f'{tag}/keys'])
    for k in sd:
        assert np.array_equal(sd[k].numpy(), g[f'{tag}/sd/{k}']), k     # same seeded init as the reference
    gen = gen.cuda()
    img = gen(cu(g[f'{tag}/embed']))[0]
    np.testing.assert_allclose(img.detach().cpu().numpy(), g[f'{tag}/img'], rtol=1e-5, atol=2e-6)
    class A: loss_type = 'L1'
    loss = orn.utils.loss_fn(img, cu(g[f'{tag}/target']), A)
    assert abs(loss.item() - float(g[f'{tag}/loss_L1'])) < 1e-6
    loss.backward()
    for k, p in gen.named_parameters():
        r = g[f'{tag}/grad/{k}']
        np.testing.assert_allclose(p.grad.cpu().numpy(), r, rtol=2e-4, atol=2e-5 * max(np.abs(r).max(), 1e-6), err_msg=k)
    if bt == 'ERB':
        for layer in gen.layers:
            layer.switch_to_deploy()
        assert list(gen.state_dict().keys()) == list(g[f'{tag}/deploy_keys'])
        with torch.no_grad():
            img_d = gen(cu(g[f'{tag}/embed']))[0]
        np.testing.assert_allclose(img_d.cpu().numpy(), g[f'{tag}/deploy_img'], rtol=1e-5, atol=2e-6)


def _make_720p(orn, bt='ERB'):
    torch.manual_seed(1)
    return orn.model.Generator(embed_length=80, stem_dim_num='512_1', fc_hw_dim='9_16_26', expansion=1, num_blocks=1,
                               norm='none', act='swish', bias=True, reduction=2, conv_type='conv',
                               stride_list=[5, 2, 2, 2, 2], sin_res=True, lower_width=96, sigmoid=False, deploy=False,
                               branch_type=bt)


def test_720p_forward_golden(orn, golden):
    """BASELINE config 2 (Bunny 720p ERB 9_16_26): decoder output vs the reference, frame for frame."""
    g = golden('generator')
    gen = _make_720p(orn).cuda()
    assert sum(p.numel() for p in gen.parameters()) == 7576025
    pe = orn.utils.PositionalEncoding('1.25_40')
    with torch.no_grad():
        for k in (0, 37):
            img = gen(pe(torch.tensor([k / 132.0])))[0].cpu().numpy()
            assert img.shape == (1, 3, 720, 1280)
            np.testing.assert_allclose(img[0, :, 352:368, 632:648], g[f'p720/frame{k}/crop'], rtol=0, atol=2e-5)
            np.testing.assert_allclose(img[0, :, :8, :8], g[f'p720/frame{k}/corner'], rtol=0, atol=2e-5)
            ms = g[f'p720/frame{k}/mean_std']
            assert abs(img.astype(np.float64).mean() - ms[0]) < 2e-6 and abs(img.astype(np.float64).std() - ms[1]) < 2e-6
            np.testing.assert_allclose(img[0].astype(np.float64).mean(axis=(0, 2)), g[f'p720/frame{k}/row_means'], atol=5e-6)


# ---------------------------------------------------------------- A11 engine ------------------
@pytest.mark.parametrize('bt', ['ERB', 'NeRV_vanilla'])
@pytest.mark.parametrize('graph', [False, True])
def test_engine_matches_oracle_training(orn, bt, graph):
    """3 optimiser steps of the native engine (Fusion6, Adam b1=0.5, schedule-driven frame/lr) vs the
    CPU oracle's autograd + Adam on the same seeded model and synthetic frames."""
    from oracle import cpu_ref
    fc, strides = '3_4_8', [2, 2, 2]
    torch.manual_seed(1)
    gen = orn.model.Generator(embed_length=80, stem_dim_num='32_1', fc_hw_dim=fc, expansion=1, num_blocks=1, norm='none',
                              act='swish', bias=True, reduction=2, conv_type='conv', stride_list=strides, sin_res=True,
                              lower_width=8, sigmoid=False, deploy=False, branch_type=bt)
    sd = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    frames = cpu_ref.synthetic_video(5, 24, 32, seed=3)
    pos = torch.tensor([k / 5.0 for k in range(5)], dtype=torch.float32)
    embeds = cpu_ref.positional_encoding(pos, 1.25, 40)
    eng = orn.engine.TrainEngine(gen, loss_type='Fusion6', beta=0.5)
    eng.set_video(frames, embeds)
    entries = [(3, 1, 5e-4), (0, 2, 4e-4), (4, 3, 3e-4)]
    eng.set_schedule(entries)
    eng.run(3, graph=graph)
    torch.cuda.synchronize()
    st = eng.stats(3).numpy()
    am = {k: torch.zeros_like(v) for k, v in sd.items()}
    av = {k: torch.zeros_like(v) for k, v in sd.items()}
    for i, (f, step, lr) in enumerate(entries):
        loss, psnr, _ = cpu_ref.train_step(sd, am, av, step, lr, embeds[f:f + 1], frames[f:f + 1], fc, strides, bt, 'Fusion6', 0.5)
        assert abs(st[i, 0] - loss.item()) < 5e-5, (i, st[i], loss.item())
        assert abs(st[i, 4] - psnr.item()) < 2e-3
        assert st[i, 6] == f and st[i, 7] == step and abs(st[i, 5] - lr) < 1e-9
    new_sd = gen.state_dict()
    for k in sd:
        # Adam normalises the update to ~lr per element, so compare in units of lr
        diff = (new_sd[k].cpu() - sd[k]).abs().max().item()
        assert diff < 0.75 * 5e-4, (k, diff)
    # the big tensors must agree far better than that on average
    k = 'stem.2.weight'
    assert (new_sd[k].cpu() - sd[k]).abs().mean().item() < 2e-5
    # decode path == eager forward of the same (arena-backed) module
    with torch.no_grad():
        img_e = eng.decode(embeds[2])
        img_m = gen(embeds[2:3].cuda())[0]
    assert torch.allclose(img_e, img_m, rtol=0, atol=1e-6)


_ORACLE_STEP = {}
_GEO = {'720p': ('9_16_26', [5, 2, 2, 2, 2], (720, 1280)),          # BASELINE configs 1 / 2
        '1080p': ('9_16_48', [5, 3, 2, 2, 2], (1080, 1920))}        # BASELINE config 3


def _make_full(orn, geo, bt):
    torch.manual_seed(1)
    fc, strides, _ = _GEO[geo]
    return orn.model.Generator(embed_length=80, stem_dim_num='512_1', fc_hw_dim=fc, expansion=1, num_blocks=1, norm='none', act='swish',
                               bias=True, reduction=2, conv_type='conv', stride_list=strides, sin_res=True, lower_width=96,
                               sigmoid=False, deploy=False, branch_type=bt)


def _oracle_full_step(geo='720p', bt='ERB'):
    """One Fusion6 training step at FULL size on the CPU oracle (autograd), computed once per session and (geometry, branch type)."""
    key = (geo, bt)
    if key not in _ORACLE_STEP:
        from oracle import cpu_ref
        import orn_amd
        fc, strides, (h, w) = _GEO[geo]
        gen = _make_full(orn_amd, geo, bt)
        sd = {k: v.detach().clone() for k, v in gen.state_dict().items()}
        frames = cpu_ref.synthetic_video(2, h, w, seed=11)
        embeds = cpu_ref.positional_encoding(torch.tensor([0.0, 0.5]), 1.25, 40)
        am = {k: torch.zeros_like(v) for k, v in sd.items()}
        av = {k: torch.zeros_like(v) for k, v in sd.items()}
        loss, psnr, ref = cpu_ref.train_step({k: v.clone() for k, v in sd.items()}, am, av, 1, 0.0, embeds[1:2], frames[1:2], fc,
                                             strides, bt, 'Fusion6', 0.5)
        _ORACLE_STEP[key] = dict(sd=sd, frames=frames, embeds=embeds, loss=loss.item(), psnr=psnr.item(), ref=ref)
    return _ORACLE_STEP[key]


def _check_full_step(orn, geo, bt, prec, n_grads):
    o = _oracle_full_step(geo, bt)
    gen = _make_full(orn, geo, bt)
    gen.load_state_dict(o['sd'])
    eng = orn.engine.TrainEngine(gen, loss_type='Fusion6', beta=0.5, precision=prec)
    eng.set_video(o['frames'], o['embeds'])
    eng.set_schedule([(1, 1, 0.0)])
    eng.run(1, graph=(prec != 'fp32'))
    torch.cuda.synchronize()
    st = eng.stats(1)[0].numpy()
    grads = {k: eng.grads[off:off + n].clone().cpu() for k, (off, n) in eng.layout.items()}
    ref = o['ref']
    tol_loss, tol_psnr, tol_g = {'fp32': (5e-5, 1e-3, 2e-3), 'fp16': (3e-4, 0.01, 1e-2), 'bf16': (2e-3, 0.05, 5e-2)}[prec]
    assert abs(st[0] - o['loss']) <= tol_loss * abs(o['loss']), (st[0], o['loss'])
    assert abs(st[4] - o['psnr']) < tol_psnr, (st[4], o['psnr'])
    assert len(ref) == n_grads
    rel = sorted(((float((grads[k] - ref[k].flatten()).norm() / (ref[k].norm() + 1e-30)), k) for k in ref), reverse=True)
    assert rel[0][0] < tol_g, rel[:8]
    assert eng.scale_state()['skipped'] == 0


@pytest.mark.parametrize('prec', ['fp32', 'fp16', 'bf16'])
def test_720p_gradients_vs_oracle(orn, prec):
    """BASELINE config 2 at FULL size: loss, PSNR and every one of the 51 gradient tensors of one Fusion6 training
    step of the engine -- in its fp32 mode AND in the 16-bit modes bench.py's headline runs in -- against the CPU ORACLE's
    autograd on the same seeded model and frame (lr 0, so the parameters stay put).  Tolerances, relative L2 per tensor:
    fp32 2e-3 (accumulation order: MFMA split-K vs ATen); fp16 1e-2 (11-bit activations / weights, fp32 accumulate);
    bf16 5e-2 (8-bit).  Loss within 5e-5 / 3e-4 / 2e-3 relative, PSNR of the prediction within 1e-3 / 0.01 / 0.05 dB."""
    _check_full_step(orn, '720p', 'ERB', prec, 51)


@pytest.mark.parametrize('prec', ['fp32', 'fp16'])
def test_720p_vanilla_gradients_vs_oracle(orn, prec):
    """BASELINE config 1's geometry (branch_type=NeRV_vanilla, 9_16_26, 720p; model.py:320-322,521-523) at FULL size against the
    CPU oracle: 16 gradient tensors.  Vanilla's 16-bit operand copies come from the prep launch, not from the merge epilogue as
    ERB's do, and its large launches (>= 128 / >= 400 pixel tiles, the slabbed wgrad) are otherwise only run by ERB models."""
    _check_full_step(orn, '720p', 'NeRV_vanilla', prec, 16)


def test_1080p_fp32_gradients_vs_oracle(orn):
    """BASELINE config 3's geometry (ERB, 9_16_48, strides 5 3 2 2 2, 1920x1080) at FULL size: the fp32 engine's loss, PSNR and 51
    gradient tensors against the CPU oracle (the 16-bit engines are compared with the fp32 engine at this size in test_gpu_bf16)."""
    _check_full_step(orn, '1080p', 'ERB', 'fp32', 51)


def test_engine_merge_is_bit_exact(orn):
    """The engine's grouped forward merge (all layers in two launches, bias folded into the S launch) against the per-op
    merge (itself bit-exact vs oracle/merge_ref.c): identical bits for every layer of the 720p model, including the
    26-channel layers with their K tails and edge tiles."""
    torch.manual_seed(3)
    gen = _make_720p(orn)
    eng = orn.engine.TrainEngine(gen, loss_type='L2', beta=0.5, precision='fp16')
    emb = torch.zeros(1, 80)
    eng.decode(emb[0])
    for li, blk in enumerate(gen.layers):
        wf_e, bf_e = eng.engine_fused_kernel(li)
        with torch.no_grad():
            wf_o, bf_o = blk.get_equivalent_kernel_bias()
        assert torch.equal(wf_e, wf_o), (li, float((wf_e - wf_o).abs().max()))
        assert torch.equal(bf_e, bf_o), li


def test_deploy_checkpoint_round_trip(orn, golden, tmp_path):
    """N1: a reference *_deploy.pth decodes frame-for-frame; our deploy export of a trained ERB model
    equals the reference's switch_to_deploy result and decodes identically through the engine's decode path."""
    from orn_amd import checkpoint
    g = golden('generator')

    def mk():
        torch.manual_seed(1)
        return orn.model.Generator(embed_length=80, stem_dim_num='32_1', fc_hw_dim='3_4_8', expansion=1, num_blocks=1,
                                   norm='none', act='swish', bias=True, reduction=2, conv_type='conv', stride_list=[2, 2],
                                   sin_res=True, lower_width=8, sigmoid=False, deploy=False, branch_type='ERB')
    embed = cu(g['tiny_ERB/embed'])
    # (1) reference deploy file -> mirror -> same image as the reference produced
    dsd = {str(k): T(g[f'tiny_ERB/deploy_sd/{k}']) for k in g['tiny_ERB/deploy_keys']}
    path = tmp_path / 'model_latest_deploy.pth'
    torch.save({'epoch': 1, 'state_dict': dsd}, path)
    gen = mk()
    assert checkpoint.load_into(gen, checkpoint.load_state_dict_file(str(path))) == 'deploy'
    gen = gen.cuda()
    with torch.no_grad():
        img = gen(embed)[0]
    np.testing.assert_allclose(img.cpu().numpy(), g['tiny_ERB/deploy_img'], rtol=1e-5, atol=2e-6)
    # (2) our export of the train-time model == the reference's deploy state dict (merge within 1e-7)
    gen_t = mk().cuda()
    exp = checkpoint.deploy_state_dict(gen_t)
    assert list(exp.keys()) == [str(k) for k in g['tiny_ERB/deploy_keys']]
    for k in exp:
        assert np.max(np.abs(exp[k].numpy() - g[f'tiny_ERB/deploy_sd/{k}'])) <= 1e-7, k
    # (3) the engine's decode path on the deploy model == eager forward
    eng = orn.engine.TrainEngine(gen, loss_type='L2', beta=0.5)
    with torch.no_grad():
        assert torch.allclose(eng.decode(embed[0]), img, rtol=0, atol=1e-6)
