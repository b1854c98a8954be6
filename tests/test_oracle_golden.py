"""Pin the CPU oracle (oracle/) against the golden vectors captured from the reference
(tools/make_golden.py).  CPU only."""
import hashlib
import math

import numpy as np
import pytest
import torch

from oracle import c_oracle, cpu_ref
from helpers import ERB_KEYS, erb_inputs

T = torch.from_numpy


def _sd(g, prefix):
    keys = [k for k in g.files if k.startswith(prefix)]
    return {k[len(prefix):]: T(g[k]) for k in keys}


# ---------------------------------------------------------------- merge (A3) -------------------
@pytest.mark.parametrize('tag,C,O', [('C6_O16', 6, 16), ('C26_O52', 26, 52)])
def test_merge_small_torch_restatement_is_bit_identical(golden, tag, C, O):
    g = golden('merge')
    w = erb_inputs(C, O, int(g[f'{tag}/seed'][0]))
    for k in ERB_KEYS:                       # the seeded generator reproduces the stored inputs
        assert np.array_equal(w[k].numpy(), g[f'{tag}/in/{k}'])
    wf, bf = cpu_ref.erb_merge(*[w[k] for k in ERB_KEYS])
    assert np.array_equal(wf.numpy(), g[f'{tag}/Wf'])        # same ATen ops -> bit identical
    assert np.array_equal(bf.numpy(), g[f'{tag}/bf'])


@pytest.mark.parametrize('tag,C,O', [('C6_O16', 6, 16), ('C26_O52', 26, 52)])
def test_merge_small_ordered_fma_restatement(golden, tag, C, O):
    """The specified-order fmaf restatement (what the HIP kernel must match bit for bit) vs the
    reference's result: <= 1e-7 abs (1-2 ulp at these magnitudes); bias bit-exact."""
    g = golden('merge')
    w = {k: g[f'{tag}/in/{k}'] for k in ERB_KEYS}
    wf, bf, Tm = c_oracle.merge_fwd(*[w[k] for k in ERB_KEYS])
    assert np.array_equal(bf, g[f'{tag}/bf'])
    assert np.max(np.abs(wf - g[f'{tag}/Wf'])) <= 1e-7
    wf_np, bf_np, T_np = cpu_ref.erb_merge_ordered_np(*[w[k] for k in ERB_KEYS])
    assert np.array_equal(wf_np, wf) and np.array_equal(T_np, Tm) and np.array_equal(bf_np, bf)


@pytest.mark.parametrize('C,O,seed', [(26, 650, 21), (26, 384, 22), (96, 384, 23), (48, 1200, 24), (48, 864, 25)])
def test_merge_real_shapes(golden, C, O, seed):
    g = golden('merge')
    tag = f'real_C{C}_O{O}'
    assert int(g[f'{tag}/seed'][0]) == seed
    w = erb_inputs(C, O, seed)
    sha = hashlib.sha256(w['rbr_1x1_3x3_1x1_branch_3x3.weight'].numpy().tobytes()).digest()
    assert np.array_equal(np.frombuffer(sha, np.uint8), g[f'{tag}/in_sha_w2'])
    wf, bf, _ = c_oracle.merge_fwd(*[w[k].numpy() for k in ERB_KEYS])
    idx = g[f'{tag}/idx']
    assert np.array_equal(bf, g[f'{tag}/bf'])
    assert np.max(np.abs(wf.reshape(-1)[idx] - g[f'{tag}/Wf_samples'])) <= 1.5e-7
    s, sa = g[f'{tag}/Wf_sum']
    assert abs(wf.astype(np.float64).sum() - s) <= 1e-6 * sa
    assert abs(np.abs(wf.astype(np.float64)).sum() - sa) <= 1e-6 * sa


@pytest.mark.parametrize('tag,C,O', [('C6_O16', 6, 16), ('C26_O52', 26, 52)])
def test_merge_backward_closed_form(golden, tag, C, O):
    """A3 backward closed forms (and the C fmaf restatement of them) vs the reference's autograd."""
    g = golden('merge')
    w = {k: T(g[f'{tag}/in/{k}']) for k in ERB_KEYS}
    G, dbf = T(g[f'{tag}/G']), T(g[f'{tag}/dbf'])
    grads = cpu_ref.erb_merge_backward_closed_form(
        G, dbf, w['rbr_1x1_3x3_1x1_branch_1x1_1.weight'], w['rbr_1x1_3x3_1x1_branch_3x3.weight'],
        w['rbr_1x1_3x3_1x1_branch_1x1_2.weight'])
    for k in ERB_KEYS:
        ref = g[f'{tag}/grad/{k}']
        assert grads[k].shape == ref.shape
        np.testing.assert_allclose(grads[k].numpy(), ref, rtol=2e-5, atol=2e-6, err_msg=k)
    _, _, Tm = c_oracle.merge_fwd(*[w[k].numpy() for k in ERB_KEYS])
    cb = c_oracle.merge_bwd(G.numpy(), w['rbr_1x1_3x3_1x1_branch_1x1_1.weight'].numpy(),
                            w['rbr_1x1_3x3_1x1_branch_3x3.weight'].numpy(),
                            w['rbr_1x1_3x3_1x1_branch_1x1_2.weight'].numpy(), Tm)
    np.testing.assert_allclose(cb['dW1'], g[f'{tag}/grad/rbr_1x1_3x3_1x1_branch_1x1_1.weight'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(cb['dW2'], g[f'{tag}/grad/rbr_1x1_3x3_1x1_branch_3x3.weight'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(cb['dW3'], g[f'{tag}/grad/rbr_1x1_3x3_1x1_branch_1x1_2.weight'], rtol=2e-5, atol=2e-6)


# ---------------------------------------------------------------- block (A4) -------------------
@pytest.mark.parametrize('s', [2, 3, 5])
def test_block_forward_backward(golden, s):
    g = golden('block')
    tag = f's{s}'
    x = T(g[f'{tag}/x']).requires_grad_(True)
    da = T(g[f'{tag}/da'])
    w = {k: T(g[f'{tag}/erb/in/{k}']).requires_grad_(True) for k in ERB_KEYS}
    wf, bf = cpu_ref.erb_merge(*[w[k] for k in ERB_KEYS])
    a = cpu_ref.block_forward(x, wf, bf, s)
    assert np.array_equal(a.detach().numpy(), g[f'{tag}/erb/a'])
    # deploy forward == train forward, bit-identical (model.py:414-432)
    assert np.array_equal(g[f'{tag}/deploy/a'], g[f'{tag}/erb/a'])
    assert np.array_equal(wf.detach().numpy(), g[f'{tag}/deploy/weight'])
    assert list(g[f'{tag}/deploy/keys']) == ['rbr_reparam.bias', 'rbr_reparam.weight']
    (a * da).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g[f'{tag}/erb/dx'], rtol=1e-5, atol=1e-6)
    for k in ERB_KEYS:
        np.testing.assert_allclose(w[k].grad.numpy(), g[f'{tag}/erb/grad/{k}'], rtol=1e-5, atol=1e-6, err_msg=k)
    # PixelShuffle index identity: z[n, h*s+i, w*s+j] = y[n*s*s + i*s + j, h, w]
    y = torch.nn.functional.conv2d(x.detach(), wf.detach(), bf.detach(), padding=1)
    z = torch.nn.functional.pixel_shuffle(y, s)
    n_, i_, j_, h_, w_ = 1, s - 1, 0, 2, 3
    assert z[0, n_, h_ * s + i_, w_ * s + j_] == y[0, n_ * s * s + i_ * s + j_, h_, w_]


# ---------------------------------------------------------------- generator (A2,A5,A6) ---------
@pytest.mark.parametrize('bt', ['ERB', 'NeRV_vanilla'])
def test_tiny_generator(golden, bt):
    g = golden('generator')
    tag = f'tiny_{bt}'
    sd = {k: T(g[f'{tag}/sd/{k}']).requires_grad_(True) for k in g[f'{tag}/keys']}
    embed, target = T(g[f'{tag}/embed']), T(g[f'{tag}/target'])
    img = cpu_ref.generator_forward(sd, embed, '3_4_8', [2, 2], bt)[0]
    assert np.array_equal(img.detach().numpy(), g[f'{tag}/img'])
    loss = cpu_ref.loss_fn(img, target, 'L1')
    assert np.array_equal(loss.detach().numpy(), g[f'{tag}/loss_L1'])
    np.testing.assert_array_equal(cpu_ref.psnr_fn([img], [target]).numpy(), g[f'{tag}/psnr'])
    loss.backward()
    for k in sd:
        np.testing.assert_allclose(sd[k].grad.numpy(), g[f'{tag}/grad/{k}'], rtol=1e-5, atol=1e-7, err_msg=k)
    # our initialiser reproduces the reference's seeded init (main_train.py:162; model.py:572-609)
    mine = cpu_ref.init_state_dict(80, '32_1', '3_4_8', [2, 2], 1, 2, 8, bt, seed=1)
    assert list(mine.keys()) == list(g[f'{tag}/keys'])
    for k in mine:
        assert np.array_equal(mine[k].numpy(), g[f'{tag}/sd/{k}']), k
    if bt == 'ERB':
        dsd = {k: T(g[f'{tag}/deploy_sd/{k}']) for k in g[f'{tag}/deploy_keys']}
        assert [str(k) for k in dsd if str(k).startswith('layers.')] == ['layers.0.rbr_reparam.weight', 'layers.0.rbr_reparam.bias',
                                                     'layers.1.rbr_reparam.weight', 'layers.1.rbr_reparam.bias']
        img_d = cpu_ref.generator_forward(dsd, embed, '3_4_8', [2, 2], 'deploy')[0]
        assert np.array_equal(img_d.numpy(), g[f'{tag}/deploy_img'])
        assert np.array_equal(g[f'{tag}/deploy_img'], g[f'{tag}/img'])


def test_720p_generator_init_and_forward(golden):
    """BASELINE config 2 (README.md:56-61): seeded init + forward of two frames vs the reference."""
    g = golden('generator')
    sd = cpu_ref.init_state_dict(80, '512_1', '9_16_26', [5, 2, 2, 2, 2], 1, 2, 96, 'ERB', seed=1)
    assert list(sd.keys()) == list(g['p720/keys'])
    assert sum(v.numel() for v in sd.values()) == int(g['p720/n_params'][0]) == 7576025
    for i, (k, v) in enumerate(sd.items()):
        assert str(tuple(v.shape)) == g['p720/shapes'][i]
        assert np.array_equal(np.resize(v.numpy().reshape(-1)[:8], 8), g['p720/param_first8'][i]), k
    torch.set_num_threads(8)
    with torch.no_grad():
        for k in (0, 37):
            embed = cpu_ref.positional_encoding(torch.tensor([k / 132.0], dtype=torch.float32), 1.25, 40)
            im = cpu_ref.generator_forward(sd, embed, '9_16_26', [5, 2, 2, 2, 2], 'ERB')[0].numpy()
            assert im.shape == (1, 3, 720, 1280)
            np.testing.assert_allclose(im[0, :, 352:368, 632:648], g[f'p720/frame{k}/crop'], rtol=0, atol=1e-6)
            np.testing.assert_allclose(im[0, :, :8, :8], g[f'p720/frame{k}/corner'], rtol=0, atol=1e-6)
            ms = g[f'p720/frame{k}/mean_std']
            assert abs(im.astype(np.float64).mean() - ms[0]) < 1e-7 and abs(im.astype(np.float64).std() - ms[1]) < 1e-7


# ---------------------------------------------------------------- utils (A1,A8,A10) ------------
def test_positional_encoding(golden):
    g = golden('utils')
    pos = T(g['pe/pos'])
    assert np.array_equal(pos.numpy(), np.array([float(k) / 132 for k in range(132)], dtype=np.float32))
    out = cpu_ref.positional_encoding(pos, 1.25, 40)
    assert out.shape == (132, 80)
    assert np.array_equal(out.numpy(), g['pe/batched'])
    assert np.array_equal(cpu_ref.positional_encoding(pos[1:2], 1.25, 40).numpy(), g['pe/single_1'])
    assert abs(float(out[1, 0]) - 0.02379770018160343) < 1e-9
    # explicit fp32 argument restatement + accurate sin/cos: <= 1 ulp (6e-8 abs) of the reference
    arg = cpu_ref.pe_arguments(pos.numpy(), 1.25, 40).astype(np.float64)
    ref = g['pe/batched'].astype(np.float64)
    assert np.max(np.abs(np.sin(arg) - ref[:, 0::2])) <= 6.1e-8
    assert np.max(np.abs(np.cos(arg) - ref[:, 1::2])) <= 6.1e-8


def test_adjust_lr_table(golden):
    g = golden('utils')
    for kind, e, it, val in g['lr/table']:
        got = cpu_ref.adjust_lr_value(int(e), int(it), 132, 5e-4, 300, 60, 'cosine' if kind == 0 else 'const')
        assert got == val, (kind, e, it)
    assert cpu_ref.adjust_lr_value(0, 0, 132, 5e-4, 300, 60) == pytest.approx(5e-5)
    assert cpu_ref.adjust_lr_value(60, 0, 132, 5e-4, 300, 60) == pytest.approx(5e-4)


def test_psnr_and_plain_losses(golden):
    g = golden('utils')
    for i in range(3):
        a, b = T(g[f'psnr/{i}/a']), T(g[f'psnr/{i}/b'])
        assert np.array_equal(cpu_ref.psnr_fn([a], [b]).numpy(), g[f'psnr/{i}/out'])
    a, b = T(g['psnr/0/a']), T(g['psnr/0/b'])
    for lt in ('L2', 'L1', 'Fusion7', 'Fusion8'):
        assert np.array_equal(cpu_ref.loss_fn(a, b, lt).numpy(), g[f'loss/{lt}'])


# ---------------------------------------------------------------- SSIM (parity unpinned) -------
def test_ssim_definition_level():
    """pytorch_msssim is absent (parity unpinned): cross-check the restatement against an
    independent SciPy implementation of the published definition and basic identities."""
    from scipy.ndimage import correlate1d
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(1, 3, 40, 52, generator=gen, dtype=torch.float64)
    y = (x + 0.1 * torch.randn(1, 3, 40, 52, generator=gen, dtype=torch.float64)).clamp(0, 1)
    assert float(cpu_ref.ssim(x, x)) == pytest.approx(1.0, abs=1e-12)
    win = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    win /= win.sum()

    def filt(a):
        a = correlate1d(a, win, axis=-2, mode='constant')[..., 5:-5, :]
        return correlate1d(a, win, axis=-1, mode='constant')[..., :, 5:-5]
    xn, yn = x.numpy(), y.numpy()
    mu1, mu2 = filt(xn), filt(yn)
    s1, s2, s12 = filt(xn * xn) - mu1 ** 2, filt(yn * yn) - mu2 ** 2, filt(xn * yn) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    ref = (((2 * mu1 * mu2 + c1) / (mu1 ** 2 + mu2 ** 2 + c1)) * ((2 * s12 + c2) / (s1 + s2 + c2))).mean()
    assert float(cpu_ref.ssim(x, y)) == pytest.approx(ref, abs=1e-12)
    ms = float(cpu_ref.ms_ssim(torch.rand(1, 3, 176, 192, generator=gen, dtype=torch.float64),
                               torch.rand(1, 3, 176, 192, generator=gen, dtype=torch.float64)))
    assert 0.0 <= ms <= 1.0
    xx = torch.rand(1, 3, 176, 192, generator=gen, dtype=torch.float64)
    assert float(cpu_ref.ms_ssim(xx, xx)) == pytest.approx(1.0, abs=1e-9)


def test_fusion6_gradient_closed_form():
    gen = torch.Generator().manual_seed(4)
    p = torch.rand(1, 3, 30, 37, generator=gen, dtype=torch.float64).requires_grad_(True)
    t = torch.rand(1, 3, 30, 37, generator=gen, dtype=torch.float64)
    cpu_ref.loss_fn(p, t, 'Fusion6').backward()
    got = cpu_ref.fusion6_grad_closed_form(p.detach(), t)
    assert torch.max(torch.abs(got - p.grad)) < 1e-15


def test_adam_matches_torch_optim():
    gen = torch.Generator().manual_seed(6)
    p0 = torch.randn(257, generator=gen)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p_ref], lr=3e-4, betas=(0.5, 0.999))
    p, m, v = p0.clone(), torch.zeros(257), torch.zeros(257)
    for step in range(1, 6):
        gr = torch.randn(257, generator=gen)
        p_ref.grad = gr.clone()
        for grp in opt.param_groups:
            grp['lr'] = 3e-4 * step
        opt.step()
        cpu_ref.adam_step(p, gr, m, v, step, 3e-4 * step)
        np.testing.assert_allclose(p.numpy(), p_ref.detach().numpy(), rtol=3e-7, atol=1e-7)  # torch>=2 uses lerp_: 1-ulp differences
