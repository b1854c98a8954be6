"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/orn.h declares,
the engine descriptor / arena layout / schedule encoding, and the LR schedule vs the golden table."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def orn():
    import orn_amd
    from orn_amd import _build
    _build.build()
    return orn_amd


def test_library_exports_every_declared_symbol(orn):
    hdr = open(os.path.join(ROOT, 'include', 'orn.h')).read()
    declared = set(re.findall(r'\b(orn_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'orn_layer_desc', 'orn_engine_desc', 'orn_step_sched'}
    L = orn._lib.lib()                       # resolves every name in _lib._SIGS (raises if one is missing)
    assert declared == set(orn._lib.EXPORTS), declared ^ set(orn._lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name)
    assert L.orn_version() == 120
    # ... and nothing else: the dynamic symbol table is the C ABI (orn.h + the probe-only orn_debug.h), no kernel stubs
    import subprocess
    dbg = set(re.findall(r'\b(orn_[a-z0-9_]+)\s*\(', open(os.path.join(ROOT, 'include', 'orn_debug.h')).read()))
    out = subprocess.run(['nm', '-D', '--defined-only', orn._lib.lib_path()], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert declared <= exported and exported <= declared | dbg, exported ^ declared
    # argument errors come back as codes + text, not exceptions across the ABI (no GPU needed)
    assert L.orn_pe_fwd(None, 0, None, 0, None, None) == -1
    assert 'pe_fwd' in orn._lib.last_error()
    assert L.orn_engine_ws_bytes(None) == 0


def test_missing_library_fails_loudly(orn, monkeypatch):
    monkeypatch.setattr(orn._lib, '_lib', None)
    monkeypatch.setattr(orn._lib, 'lib_path', lambda: '/nonexistent/liborn.so')
    with pytest.raises(orn.OrnError):
        orn._lib.lib()


def test_cpu_tensors_are_rejected(orn):
    from orn_amd import ops
    with pytest.raises(orn.OrnError):
        ops.ConvPsSiluFn.apply(torch.zeros(1, 4, 4, 4), torch.zeros(16, 4, 3, 3), torch.zeros(16), 2)
    with pytest.raises(orn.OrnError):
        ops.loss_stats(torch.zeros(1, 3, 16, 16), torch.zeros(1, 3, 16, 16), 'L2')


def _gen(orn, bt='ERB', deploy=False):
    from orn_amd import model
    torch.manual_seed(1)
    return model.Generator(embed_length=80, stem_dim_num='512_1', fc_hw_dim='9_16_26', expansion=1, num_blocks=1,
                           norm='none', act='swish', bias=True, reduction=2, conv_type='conv', stride_list=[5, 2, 2, 2, 2],
                           sin_res=True, lower_width=96, sigmoid=False, deploy=deploy, branch_type=bt)


def test_generator_mirror_matches_reference_layout(orn, golden):
    g = golden('generator')
    gen = _gen(orn)
    sd = gen.state_dict()
    assert list(sd.keys()) == list(g['p720/keys'])                   # 51 tensors, reference key order
    assert sum(p.numel() for p in gen.parameters()) == 7576025
    for i, (k, v) in enumerate(sd.items()):                          # same seeded init as main_train.py:162
        assert str(tuple(v.shape)) == g['p720/shapes'][i]
        assert np.array_equal(np.resize(v.numpy().reshape(-1)[:8], 8), g['p720/param_first8'][i]), k
    assert hasattr(gen.layers[0], 'rbr_3x3_branch') and gen.stem[0].in_features == 80
    dep = _gen(orn, deploy=True)
    assert [k for k in dep.state_dict() if k.startswith('layers.0')] == ['layers.0.rbr_reparam.weight', 'layers.0.rbr_reparam.bias']
    assert sum(p.numel() for p in dep.parameters()) == 3201905
    with pytest.raises(NotImplementedError):
        from orn_amd import model
        model.NeRVBlock(ngf=4, new_ngf=4, stride=2, bias=True, norm='none', act='swish', deploy=False, conv_type='conv',
                        branch_type='DBB')


def test_engine_descriptor_and_arena(orn):
    from orn_amd import engine
    gen = _gen(orn)
    named = [(k, tuple(p.shape)) for k, p in gen.named_parameters()]
    layout, total = engine.arena_layout(named)
    assert total % 64 == 0 and all(off % 64 == 0 for off, _ in layout.values())
    offs = sorted(layout.values())
    for (o0, n0), (o1, _) in zip(offs, offs[1:]):
        assert o0 + n0 <= o1                                          # no overlap
    d = engine.build_desc(gen, layout, total, 'Fusion6', 0.5)
    assert (d.n_layers, d.erb, d.embed_len, d.stem_dim, d.fc_h, d.fc_w, d.fc_dim) == (5, 1, 80, 512, 9, 16, 26)
    geo = [(L.C, L.O, L.s, L.H, L.W) for L in list(d.layer)[:5]]
    assert geo == [(26, 650, 5, 9, 16), (26, 384, 2, 45, 80), (96, 384, 2, 90, 160), (96, 384, 2, 180, 320), (96, 384, 2, 360, 640)]
    assert d.layer[4].w3 == layout['layers.4.rbr_1x1_3x3_1x1_branch_1x1_2.weight'][0]
    assert d.head_w == layout['head_layers.4.weight'][0] and d.loss_type == 2 and d.beta1 == 0.5
    assert orn._lib.lib().orn_engine_ws_bytes(d) > 0                 # geometry accepted by the native side
    d.layer[2].H = 91
    assert orn._lib.lib().orn_engine_ws_bytes(d) == 0 and 'does not chain' in orn._lib.last_error()
    van = _gen(orn, 'NeRV_vanilla')
    lv, tv = engine.arena_layout([(k, tuple(p.shape)) for k, p in van.named_parameters()])
    dv = engine.build_desc(van, lv, tv)
    assert dv.erb == 0 and dv.layer[0].w3x3 == lv['layers.0.branch.weight'][0] and dv.layer[0].w1 == -1


def test_schedule_encoding(orn):
    from orn_amd import engine
    arr = engine.make_schedule([(3, 1, 5e-4), (131, 39600, 1.229e-12)])
    assert arr.dtype == np.int32 and arr.shape == (2, 4)
    assert arr[1, 0] == 131 and arr[1, 1] == 39600
    assert arr[:, 2].view(np.float32)[0] == np.float32(5e-4)


def test_lr_schedule_matches_reference(orn, golden):
    from orn_amd import utils

    class A:
        lr, epochs, warmup, lr_steps = 5e-4, 300, 60, []
    g = golden('utils')
    for kind, e, it, val in g['lr/table']:
        A.lr_type = 'cosine' if kind == 0 else 'const'
        assert utils.lr_value(int(e), int(it), 132, A) == val
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))])
    A.lr_type = 'cosine'
    assert utils.adjust_lr(opt, 30, 0, 132, A) == pytest.approx(2.75e-4) and opt.param_groups[0]['lr'] == pytest.approx(2.75e-4)
    pe = utils.PositionalEncoding('1.25_40')
    assert pe.embed_length == 80 and pe.lbase == 1.25 and pe.levels == 40


def test_checkpoint_interchange_layouts(orn, golden, tmp_path):
    """Reference-format checkpoints (train ERB and deploy key layouts) load into the mirror, safely
    (weights_only), and the deploy structure switch needs no GPU."""
    from orn_amd import checkpoint, model
    g = golden('generator')

    def mk(bt='ERB'):
        torch.manual_seed(1)
        return model.Generator(embed_length=80, stem_dim_num='32_1', fc_hw_dim='3_4_8', expansion=1, num_blocks=1, norm='none',
                               act='swish', bias=True, reduction=2, conv_type='conv', stride_list=[2, 2], sin_res=True,
                               lower_width=8, sigmoid=False, deploy=False, branch_type=bt)
    # a reference train checkpoint (state dict captured from the reference) with thop's stray buffers
    sd = {str(k): torch.from_numpy(g[f'tiny_ERB/sd/{k}']) for k in g['tiny_ERB/keys']}
    sd['layers.0.total_ops'] = torch.zeros(1)
    path = tmp_path / 'model_latest.pth'
    torch.save({'epoch': 3, 'state_dict': sd, 'train_best_psnr': torch.tensor(1.0), 'optimizer': {}}, path)
    loaded = checkpoint.load_state_dict_file(str(path))
    assert 'layers.0.total_ops' not in loaded and checkpoint.state_dict_kind(loaded) == 'ERB'
    gen = mk()
    assert checkpoint.load_into(gen, loaded) == 'ERB'
    for k in g['tiny_ERB/keys']:
        assert np.array_equal(gen.state_dict()[str(k)].numpy(), g[f'tiny_ERB/sd/{k}'])
    # a reference deploy checkpoint
    dsd = {str(k): torch.from_numpy(g[f'tiny_ERB/deploy_sd/{k}']) for k in g['tiny_ERB/deploy_keys']}
    dpath = tmp_path / 'model_latest_deploy.pth'
    torch.save({'epoch': 3, 'state_dict': dsd}, dpath)
    gen2 = mk()
    assert checkpoint.load_into(gen2, checkpoint.load_state_dict_file(str(dpath))) == 'deploy'
    assert list(gen2.state_dict().keys()) == [str(k) for k in g['tiny_ERB/deploy_keys']]
    assert all(b.deploy for b in gen2.layers)
    assert np.array_equal(gen2.layers[1].rbr_reparam.weight.detach().numpy(), g['tiny_ERB/deploy_sd/layers.1.rbr_reparam.weight'])
    # vanilla layout is recognised too; garbage is rejected
    assert checkpoint.state_dict_kind(mk('NeRV_vanilla').state_dict()) == 'NeRV_vanilla'
    with pytest.raises(ValueError):
        checkpoint.state_dict_kind({'foo': torch.zeros(1)})
    # our own save() writes the reference's dict keys
    ck = checkpoint.save(str(tmp_path / 'x' / 'model_latest.pth'), mk('NeRV_vanilla'), epoch=7)
    assert set(ck.keys()) == {'epoch', 'state_dict', 'train_best_psnr', 'train_best_msssim', 'val_best_psnr', 'val_best_msssim', 'optimizer'}


def test_quantize_per_tensor_matches_reference(orn, golden):
    """utils.py:11-67 incl. the mask->ones and zero-handling quirks, from the reference's own outputs."""
    from orn_amd import eval_utils
    g = golden('utils')
    cases = sorted({k.split('/')[1] + '/' + k.split('/')[2] for k in g.files if k.startswith('quant/')})
    assert len(cases) >= 10
    for c in cases:
        axis = int(c.split('axis')[1])
        if f'quant/{c}/raises' in g.files:
            with pytest.raises(Exception):
                eval_utils.quantize_per_tensor(torch.zeros(17), 8, axis)
            continue
        t = torch.from_numpy(g[f'quant/{c}/in'])
        q, n = eval_utils.quantize_per_tensor(t.clone(), 8, axis)
        assert np.array_equal(q.numpy(), g[f'quant/{c}/quant']), c
        assert np.array_equal(n.numpy(), g[f'quant/{c}/new']), c
    # quirk Q3: a 0/1 mask row becomes all ones
    qm, nm = eval_utils.quantize_per_tensor(torch.tensor([[1., 0., 1.], [0., 0., 0.]]), 8, 0)
    assert torch.equal(nm[0], torch.ones(3))


def test_prune_and_huffman(orn):
    from orn_amd import eval_utils
    import torch.nn.utils.prune as prune
    torch.manual_seed(0)
    a, b = torch.nn.Linear(13, 7), torch.nn.Conv2d(3, 5, 3)
    masks = eval_utils.global_l1_prune_masks({'a': a.weight, 'b': b.weight}, 0.4)
    prune.global_unstructured([(a, 'weight'), (b, 'weight')], pruning_method=prune.L1Unstructured, amount=0.4)
    assert torch.equal(masks['a'], a.weight_mask) and torch.equal(masks['b'], b.weight_mask)
    # Huffman: known answer (frequencies 5,9,12,13,16,45 -> 224 bits), single-symbol stream, and the entropy bound
    syms = [0] * 5 + [1] * 9 + [2] * 12 + [3] * 13 + [4] * 16 + [5] * 45
    assert eval_utils.huffman_total_bits(syms) == 224
    assert eval_utils.huffman_total_bits([7] * 10) == 10
    import math
    n = len(syms)
    ent = -sum(c / n * math.log2(c / n) for c in (5, 9, 12, 13, 16, 45)) * n
    assert ent <= 224 < ent + n
    sd, bits, count, hist = eval_utils.quantized_model_bits({'w': torch.randn(6, 10), 'b': torch.randn(6)}, 8, 0)
    assert count == 66 and 0 < bits <= 66 * 9 and sd['w'].shape == (6, 10) and sum(hist.values()) == 66


def test_prune_then_quantise_matches_reference_walk(orn):
    """G9 / Q3 (tests/golden/prune.npz, made by the reference's Generator + torch prune + quantize_per_tensor): deploy-mode tiny
    ERB model, global L1 prune 0.4 over the stem Linear weights and the rbr_reparam convs, then the quantisation walk of
    main_eval.py:659-669.  Pinned: the pruned state dict's key list (weight_orig / weight_mask pairs), every de-quantised
    tensor bit for bit (the masks come back as all ones: quantisation un-prunes), the number of coded entries per tensor
    (non-zero entries only) and the level histogram the Huffman table is built from."""
    from orn_amd import eval_utils
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'prune.npz'))
    dsd = {k[len('q3/deploy_sd/'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('q3/deploy_sd/')}
    keys = [str(k) for k in g['q3/keys']]
    order = [k for k in keys if not k.endswith('_mask')]
    sd = {k.replace('weight_orig', 'weight'): dsd[k.replace('weight_orig', 'weight')] for k in order}
    prunable = {k: v for k, v in sd.items()
                if k.endswith('.weight') and (k.startswith('stem.') or (k.startswith('layers.') and k.split('.')[2] == 'rbr_reparam'))}
    masks = eval_utils.global_l1_prune_masks(prunable, 0.4)
    psd = eval_utils.pruned_state_dict(sd, masks)
    assert list(psd.keys()) == keys
    new_sd, bits, count, hist = eval_utils.quantized_model_bits(psd, 8, 0)
    for k in keys:
        assert torch.equal(new_sd[k], torch.from_numpy(g[f'q3/new/{k}'])), k
        nv = int(g[f'q3/n_valid/{k}'][0])
        assert nv == int((psd[k] != 0).sum()), k
        if k.endswith('_mask'):
            assert torch.equal(new_sd[k], torch.ones_like(new_sd[k]))          # Q3
    assert count == int(g['q3/n_symbols'][0])
    ref_hist = {float(v): int(c) for v, c in zip(g['q3/level_values'], g['q3/level_counts'])}
    assert hist == ref_hist
    assert bits == eval_utils.huffman_bits_from_counts(ref_hist.values()) and count * 1 <= bits <= count * 9
    # the folded model is un-pruned (orig * ones) but quantised
    folded = eval_utils.fold_pruned(new_sd)
    assert set(folded) == set(sd) and torch.equal(folded['stem.0.weight'], new_sd['stem.0.weight_orig'])


def test_finetune_lr_matches_reference(orn):
    """Q2: the LR of the prune fine-tune (main_eval.py:474) for epochs 300..399 of a 300-epoch checkpoint, against the
    reference's adjust_lr (tests/golden/prune.npz q2): the cosine evaluated past its end with warm-up 60 EPOCHS -- main_eval
    normalises --warmup 0.2 to int(0.2 * epochs) exactly as main_train does."""
    from orn_amd import utils, main_eval, main_train
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'prune.npz'))
    args = main_train.build_parser().parse_args(['--epochs', '300', '--lr', '0.0005', '--warmup', '0.2'])     # README.md:56-61
    args.warmup = int(args.warmup * args.epochs)
    assert args.warmup == 60 and args.epochs == 300 and args.lr == 5e-4
    for epoch, it, lr in g['q2/finetune_lr']:
        mine = utils.lr_value(int(epoch) % 400, int(it), 132, args)
        assert mine == lr, (epoch, it, mine, lr)
    import inspect
    assert 'args.warmup = int(args.warmup * args.epochs)' in inspect.getsource(main_eval.main)


def _write_frames(d, n, portrait=False):
    from PIL import Image
    rng = np.random.RandomState(7)
    imgs = []
    for k in range(n):
        a = rng.randint(0, 256, size=((12, 8, 3) if portrait else (8, 12, 3)), dtype=np.uint8)
        Image.fromarray(a).save(os.path.join(d, f'f{k:04d}.png'))
        imgs.append(a)
    return imgs


def test_frame_dir_matches_customdataset(orn, tmp_path):
    """N4: FrameDir against CustomDataSet's indexing (model.py:11-70) on a temporary PNG directory: odd frame count with
    frame_gap 2 (floor(N / gap) samples, sample k = file k*gap at time k*gap / N_all), a vid_list of frame indices (subsets the
    TIME table only, model.py:40-41), ToTensor scaling, and the portrait -> landscape transpose (model.py:66-67)."""
    from orn_amd import data
    d = tmp_path / 'land'
    d.mkdir()
    imgs = _write_frames(str(d), 7)
    fd = data.FrameDir(str(d), (None,), 2)
    assert len(fd) == 3                                           # 7 // 2, not ceil
    frames, t = fd.load('cpu')
    assert frames.shape == (3, 3, 8, 12) and frames.dtype == torch.float32
    for k in range(3):
        want = torch.from_numpy(imgs[2 * k]).permute(2, 0, 1).float() / 255.0
        assert torch.equal(frames[k], want)
        assert t[k].item() == torch.tensor(float(2 * k) / 7).item()
    fd1 = data.FrameDir(str(d), (None,), 1)
    assert len(fd1) == 7 and fd1.item(6)[1] == 6.0 / 7
    # vid_list: indices into the time table; the file table is not subset (reference behaviour, kept)
    fv = data.FrameDir(str(d), [5, 1, 3], 1)
    assert len(fv) == 3
    f0, t0 = fv.item(0)
    assert t0 == 5.0 / 7 and torch.equal(f0, torch.from_numpy(imgs[0]).permute(2, 0, 1))
    assert fv.item(2)[1] == 3.0 / 7
    assert len(data.FrameDir(str(d), [5, 1, 3], 2)) == 1
    # portrait frames come back transposed to landscape
    dp = tmp_path / 'port'
    dp.mkdir()
    pimgs = _write_frames(str(dp), 2, portrait=True)
    pf, _ = data.load_png_dir(str(dp), (None,), 1, 'cpu')
    assert pf.shape == (2, 3, 8, 12)
    assert torch.equal(pf[1], (torch.from_numpy(pimgs[1]).permute(2, 0, 1).permute(0, 2, 1).float() / 255.0))
    with pytest.raises(FileNotFoundError):
        data.load_png_dir(str(tmp_path / 'land'), (None,), 8, 'cpu')


def test_per_rank_outputs_do_not_collide(orn):
    """Multi-rank CLI: the job's videos are dealt round-robin, one output directory per video."""
    from orn_amd import main_train, dist_utils as du

    class A:
        synthetic, dataset = 0, 'beauty,bosphorus,honeybee,jockey,readysetgo,shakendry,yachtride'
    vids = main_train.video_list(A, 8)
    assert len(vids) == 7
    seen = {}
    for rank in range(8):
        for v in du.shard_videos(len(vids), 8, rank):
            out = main_train.video_outf('result/uvg', vids, v)
            assert out not in seen, (out, rank, seen[out])
            seen[out] = rank
    assert len(seen) == 7 and seen['result/uvg/beauty'] == 0 and du.shard_videos(7, 8, 7) == []
    A.synthetic = 12
    sv = main_train.video_list(A, 2)
    assert sv == ['synthetic0', 'synthetic1'] and main_train.video_outf('o', sv, 1) == 'o/synthetic1'
    A.synthetic, A.dataset = 0, 'bunny'
    assert main_train.video_outf('result/b', main_train.video_list(A, 1), 0) == 'result/b'       # single video: reference layout


def test_skipped_steps_warning():
    """main_train's dead-fit detector: silent while the guard does its normal work, loud when most of an epoch was skipped."""
    from orn_amd import main_train as mt
    assert mt.skipped_steps_warning(0, 0, 132, 'fp16', 2.0 ** 20) is None
    assert mt.skipped_steps_warning(10, 14, 132, 'fp16', 2.0 ** 19) is None             # a handful of back-offs
    assert mt.skipped_steps_warning(10, 76, 132, 'fp16', 2.0 ** 19) is None             # exactly half: still quiet
    w = mt.skipped_steps_warning(10, 142, 132, 'fp16', 1.0)
    assert w and '132 of 132' in w and 'fp32' in w and 'bf16' in w
    w = mt.skipped_steps_warning(0, 100, 132, 'fp32', 1.0)
    assert w and 'bf16' not in w
