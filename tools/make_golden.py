#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the reference (read-only at /root/reference).

Runs ONLY in the build container (the reference never travels to the GPU box); the fixtures it
writes are data -- seeded inputs and the reference's outputs -- and are committed.  Nothing here is
imported by the product or by the tests.

`utils.py` does `from pytorch_msssim import ms_ssim, ssim` (utils.py:9); that package is absent from
this image, so an EMPTY placeholder module is registered purely so the import statement succeeds.
It implements nothing: SSIM / MS-SSIM outputs are never produced here and stay "parity unpinned".
"""
import hashlib
import math
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def _import_reference():
    sys.path.insert(0, REF)
    import model as ref_model                                   # noqa: E402
    stub = types.ModuleType('pytorch_msssim')

    def _absent(*a, **k):
        raise NotImplementedError('pytorch_msssim is not available in this image')
    stub.ssim = _absent
    stub.ms_ssim = _absent
    sys.modules['pytorch_msssim'] = stub
    import utils as ref_utils                                   # noqa: E402
    return ref_model, ref_utils


def _np(t):
    return t.detach().cpu().numpy().copy()


def _rand(gen, *shape, scale=1.0):
    return (torch.rand(*shape, generator=gen) * 2 - 1) * scale


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


ERB_KEYS = (
    'rbr_3x3_branch.weight', 'rbr_3x3_branch.bias', 'rbr_3x1_branch.weight', 'rbr_3x1_branch.bias',
    'rbr_1x3_branch.weight', 'rbr_1x3_branch.bias', 'rbr_1x1_3x3_1x1_branch_1x1_1.weight',
    'rbr_1x1_3x3_1x1_branch_3x3.weight', 'rbr_1x1_3x3_1x1_branch_1x1_2.weight')


def erb_inputs(C, O, seed):
    """Seeded branch weights with PyTorch-default-like magnitudes; reproduced verbatim by the tests."""
    g = torch.Generator().manual_seed(seed)
    return {
        'rbr_3x3_branch.weight': _rand(g, O, C, 3, 3, scale=1 / math.sqrt(9 * C)),
        'rbr_3x3_branch.bias': _rand(g, O, scale=1 / math.sqrt(9 * C)),
        'rbr_3x1_branch.weight': _rand(g, O, C, 3, 1, scale=1 / math.sqrt(3 * C)),
        'rbr_3x1_branch.bias': _rand(g, O, scale=1 / math.sqrt(3 * C)),
        'rbr_1x3_branch.weight': _rand(g, O, C, 1, 3, scale=1 / math.sqrt(3 * C)),
        'rbr_1x3_branch.bias': _rand(g, O, scale=1 / math.sqrt(3 * C)),
        'rbr_1x1_3x3_1x1_branch_1x1_1.weight': _rand(g, 2 * C, C, 1, 1, scale=1 / math.sqrt(C)),
        'rbr_1x1_3x3_1x1_branch_3x3.weight': _rand(g, O, 2 * C, 3, 3, scale=1 / math.sqrt(18 * C)),
        'rbr_1x1_3x3_1x1_branch_1x1_2.weight': _rand(g, O, O, 1, 1, scale=1 / math.sqrt(O)),
    }


def make_block(ref_model, C, new_ngf, s, branch_type, deploy=False):
    return ref_model.NeRVBlock(ngf=C, new_ngf=new_ngf, stride=s, bias=True, norm='none', act='swish',
                               deploy=deploy, conv_type='conv', branch_type=branch_type)


def golden_merge(ref_model):
    out = {}
    # small shapes: full tensors + autograd grads
    for (C, new_ngf, s, seed) in [(6, 4, 2, 11), (26, 13, 2, 12)]:
        O = new_ngf * s * s
        blk = make_block(ref_model, C, new_ngf, s, 'ERB')
        w = erb_inputs(C, O, seed)
        blk.load_state_dict(w)
        wf, bf = blk.get_equivalent_kernel_bias()
        g = torch.Generator().manual_seed(seed + 100)
        G = _rand(g, O, C, 3, 3)
        dbf = _rand(g, O)
        ((wf * G).sum() + (bf * dbf).sum()).backward()
        tag = f'C{C}_O{O}'
        out[f'{tag}/seed'] = np.array([seed])
        for k in ERB_KEYS:
            out[f'{tag}/in/{k}'] = _np(w[k])
            out[f'{tag}/grad/{k}'] = _np(dict(blk.named_parameters())[k].grad)
        out[f'{tag}/Wf'] = _np(wf)
        out[f'{tag}/bf'] = _np(bf)
        out[f'{tag}/G'] = _np(G)
        out[f'{tag}/dbf'] = _np(dbf)
    # real shapes (BASELINE configs 2 and 3): seeded inputs (regenerated in the test), sampled outputs
    for (C, O, seed) in [(26, 650, 21), (26, 384, 22), (96, 384, 23), (48, 1200, 24), (48, 864, 25)]:
        blk = make_block(ref_model, C, O, 1, 'ERB')          # stride 1: O = new_ngf
        w = erb_inputs(C, O, seed)
        blk.load_state_dict(w)
        with torch.no_grad():
            wf, bf = blk.get_equivalent_kernel_bias()
        wf_np, bf_np = _np(wf), _np(bf)
        idx = np.random.RandomState(seed).choice(wf_np.size, 256, replace=False)
        tag = f'real_C{C}_O{O}'
        out[f'{tag}/seed'] = np.array([seed])
        out[f'{tag}/idx'] = idx.astype(np.int64)
        out[f'{tag}/Wf_samples'] = wf_np.reshape(-1)[idx]
        out[f'{tag}/Wf_sum'] = np.array([wf_np.astype(np.float64).sum(), np.abs(wf_np.astype(np.float64)).sum()])
        out[f'{tag}/bf'] = bf_np
        out[f'{tag}/in_sha_w2'] = np.frombuffer(bytes.fromhex(sha(_np(w['rbr_1x1_3x3_1x1_branch_3x3.weight']))), np.uint8)
    np.savez_compressed(os.path.join(OUT, 'merge.npz'), **out)


def golden_block(ref_model):
    """NeRVBlock fwd/bwd: ERB, vanilla, deploy on x[1,6,5,7], s in {2,3,5} (model.py:518-567)."""
    out = {}
    C, new_ngf, H, W = 6, 4, 5, 7
    for s in (2, 3, 5):
        O = new_ngf * s * s
        g = torch.Generator().manual_seed(300 + s)
        x = _rand(g, 1, C, H, W).requires_grad_(True)
        da = _rand(g, 1, new_ngf, H * s, W * s)
        # ERB
        blk = make_block(ref_model, C, new_ngf, s, 'ERB')
        w = erb_inputs(C, O, 400 + s)
        blk.load_state_dict(w)
        a = blk(x)
        (a * da).sum().backward()
        tag = f's{s}'
        out[f'{tag}/x'] = _np(x)
        out[f'{tag}/da'] = _np(da)
        out[f'{tag}/erb/a'] = _np(a)
        out[f'{tag}/erb/dx'] = _np(x.grad)
        for k in ERB_KEYS:
            out[f'{tag}/erb/in/{k}'] = _np(w[k])
            out[f'{tag}/erb/grad/{k}'] = _np(dict(blk.named_parameters())[k].grad)
        # deploy == train forward (model.py:395-448)
        with torch.no_grad():
            wf, bf = blk.get_equivalent_kernel_bias()
        blk.switch_to_deploy()
        out[f'{tag}/deploy/keys'] = np.array(sorted(blk.state_dict().keys()))
        out[f'{tag}/deploy/weight'] = _np(blk.rbr_reparam.weight)
        out[f'{tag}/deploy/bias'] = _np(blk.rbr_reparam.bias)
        with torch.no_grad():
            out[f'{tag}/deploy/a'] = _np(blk(x))
        assert torch.equal(blk.rbr_reparam.weight, wf)
        # vanilla with the merged kernel as its weight
        van = make_block(ref_model, C, new_ngf, s, 'NeRV_vanilla')
        van.load_state_dict({'branch.weight': wf, 'branch.bias': bf})
        x2 = x.detach().clone().requires_grad_(True)
        a2 = van(x2)
        (a2 * da).sum().backward()
        out[f'{tag}/vanilla/a'] = _np(a2)
        out[f'{tag}/vanilla/dx'] = _np(x2.grad)
        out[f'{tag}/vanilla/dW'] = _np(van.branch.weight.grad)
        out[f'{tag}/vanilla/db'] = _np(van.branch.bias.grad)
    np.savez_compressed(os.path.join(OUT, 'block.npz'), **out)


def make_generator(ref_model, embed_length, stem, fc, strides, lower_width, branch_type, deploy=False,
                   expansion=1, reduction=2):
    return ref_model.Generator(embed_length=embed_length, stem_dim_num=stem, fc_hw_dim=fc, expansion=expansion,
                               num_blocks=1, norm='none', act='swish', bias=True, reduction=reduction,
                               conv_type='conv', stride_list=strides, sin_res=True, lower_width=lower_width,
                               sigmoid=False, deploy=deploy, branch_type=branch_type)


def golden_generator(ref_model, ref_utils):
    out = {}
    pe = ref_utils.PositionalEncoding('1.25_40')
    # tiny generator: fc 3_4_8, strides [2,2], lower_width 8, stem 32_1 -- full state, L1 loss grads
    for bt in ('ERB', 'NeRV_vanilla'):
        torch.manual_seed(1)
        gen = make_generator(ref_model, 80, '32_1', '3_4_8', [2, 2], 8, bt)
        sd = {k: v.detach().clone() for k, v in gen.state_dict().items()}
        pos = torch.tensor([3.0 / 7.0], dtype=torch.float32)
        embed = pe(pos)
        g = torch.Generator().manual_seed(77)
        target = torch.rand(1, 3, 12, 16, generator=g)
        img = gen(embed)[0]
        loss = torch.mean(torch.abs(img - target))
        loss.backward()
        tag = f'tiny_{bt}'
        out[f'{tag}/embed'] = _np(embed)
        out[f'{tag}/target'] = _np(target)
        out[f'{tag}/img'] = _np(img)
        out[f'{tag}/loss_L1'] = _np(loss)
        out[f'{tag}/psnr'] = _np(ref_utils.psnr_fn([img], [target]))
        out[f'{tag}/keys'] = np.array(list(sd.keys()))
        for k, v in sd.items():
            out[f'{tag}/sd/{k}'] = _np(v)
        for k, p in gen.named_parameters():
            out[f'{tag}/grad/{k}'] = _np(p.grad)
        if bt == 'ERB':
            # deploy checkpoint round trip (main_train.py:325-351)
            for layer in gen.layers:
                layer.switch_to_deploy()
            dsd = gen.state_dict()
            out[f'{tag}/deploy_keys'] = np.array(list(dsd.keys()))
            for k, v in dsd.items():
                out[f'{tag}/deploy_sd/{k}'] = _np(v)
            with torch.no_grad():
                out[f'{tag}/deploy_img'] = _np(gen(embed)[0])
    # 720p ERB, BASELINE config 2: init under manual_seed(1) (main_train.py:162), two frames
    torch.manual_seed(1)
    gen = make_generator(ref_model, 80, '512_1', '9_16_26', [5, 2, 2, 2, 2], 96, 'ERB')
    sd = gen.state_dict()
    out['p720/keys'] = np.array(list(sd.keys()))
    out['p720/shapes'] = np.array([str(tuple(v.shape)) for v in sd.values()])
    out['p720/param_sums'] = np.array([float(v.double().sum()) for v in sd.values()])
    out['p720/param_first8'] = np.stack([np.resize(_np(v).reshape(-1)[:8], 8) for v in sd.values()])
    out['p720/n_params'] = np.array([sum(p.numel() for p in gen.parameters())])
    with torch.no_grad():
        for k in (0, 37):
            embed = pe(torch.tensor([k / 132.0], dtype=torch.float32))
            img = gen(embed)[0]
            im = _np(img)
            out[f'p720/frame{k}/mean_std'] = np.array([im.astype(np.float64).mean(), im.astype(np.float64).std()])
            out[f'p720/frame{k}/crop'] = im[0, :, 352:368, 632:648].copy()
            out[f'p720/frame{k}/corner'] = im[0, :, :8, :8].copy()
            out[f'p720/frame{k}/row_means'] = im[0].astype(np.float64).mean(axis=(0, 2)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, 'generator.npz'), **out)


def golden_utils(ref_utils):
    out = {}
    pe = ref_utils.PositionalEncoding('1.25_40')
    assert pe.embed_length == 80
    pos = torch.tensor([float(k) / 132 for k in range(132)], dtype=torch.float32)   # model.py:37,68
    out['pe/pos'] = _np(pos)
    out['pe/batched'] = _np(pe(pos))
    out['pe/single_1'] = _np(pe(pos[1:2]))
    # adjust_lr (utils.py:240-259) for README config: lr 5e-4, 300 epochs, warmup int(0.2*300)
    class A:
        pass
    rows = []
    for lr_type in ('cosine', 'const'):
        a = A()
        a.lr, a.epochs, a.warmup, a.lr_type, a.lr_steps = 5e-4, 300, 60, lr_type, []
        opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))])
        for e in list(range(0, 300, 10)) + [299, 300, 301, 400]:
            for it in (0, 66, 131):
                rows.append([0 if lr_type == 'cosine' else 1, e, it, ref_utils.adjust_lr(opt, e, it, 132, a)])
    out['lr/table'] = np.array(rows, dtype=np.float64)
    # psnr_fn (utils.py:191-199)
    g = torch.Generator().manual_seed(5)
    for i in range(3):
        a_ = torch.rand(2, 3, 9, 11, generator=g)
        b_ = (a_ + 0.05 * (i + 1) * torch.randn(2, 3, 9, 11, generator=g)).clamp(0, 1)
        out[f'psnr/{i}/a'] = _np(a_)
        out[f'psnr/{i}/b'] = _np(b_)
        out[f'psnr/{i}/out'] = _np(ref_utils.psnr_fn([a_], [b_]))
    # loss_fn SSIM-free branches (utils.py:142-145,161-164)
    for lt in ('L2', 'L1', 'Fusion7', 'Fusion8'):
        a = A()
        a.loss_type = lt
        out[f'loss/{lt}'] = _np(ref_utils.loss_fn(torch.from_numpy(out['psnr/0/a']), torch.from_numpy(out['psnr/0/b']), a))
    # quantize_per_tensor (utils.py:11-67), incl. the 0/1 mask and zero-row quirks
    g = torch.Generator().manual_seed(9)
    cases = {
        'rand2d': torch.randn(6, 10, generator=g),
        'rand4d': torch.randn(4, 3, 3, 3, generator=g),
        'mask01': (torch.rand(5, 8, generator=g) > 0.4).float(),
        'withzeros': torch.randn(6, 10, generator=g) * (torch.rand(6, 10, generator=g) > 0.3).float(),
        'vec': torch.randn(17, generator=g),
    }
    for name, t in cases.items():
        for axis in (-1, 0, 1):
            if axis == 1 and t.dim() < 2:
                continue
            try:
                qt, nt = ref_utils.quantize_per_tensor(t.clone(), 8, axis)
            except Exception as exc:                             # record that the reference raises
                out[f'quant/{name}/axis{axis}/raises'] = np.array([type(exc).__name__])
                continue
            out[f'quant/{name}/axis{axis}/in'] = _np(t)
            out[f'quant/{name}/axis{axis}/quant'] = _np(qt)
            out[f'quant/{name}/axis{axis}/new'] = _np(nt)
    np.savez_compressed(os.path.join(OUT, 'utils.npz'), **out)


ERB_BRANCH_MODULES = ('rbr_3x3_branch', 'rbr_3x1_branch', 'rbr_1x3_branch', 'rbr_1x1_3x3_1x1_branch_1x1_1',
                      'rbr_1x1_3x3_1x1_branch_3x3', 'rbr_1x1_3x3_1x1_branch_1x1_2')


def golden_prune(ref_model, ref_utils):
    """G9: the prune -> fine-tune quirk Q1 and the prune -> quantise quirk Q3 of main_eval.py, produced by the
    reference's own modules (NeRVBlock, Generator, quantize_per_tensor) + torch.nn.utils.prune.  main_eval.py itself cannot be
    imported here (torchvision / thop / dahuffman are absent), so the few glue lines that select the modules
    (main_eval.py:296-340 train-mode ERB, :571-587 deploy-mode ERB) and walk the state dict (main_eval.py:659-669) are
    restated below; every number comes out of reference code."""
    import torch.nn.utils.prune as prune
    out = {}
    pe = ref_utils.PositionalEncoding('1.25_40')
    # ---- Q1: train-mode ERB, global L1 prune 0.4, three Adam steps (main_eval.py:296-350, 450-499) --------------------
    torch.manual_seed(1)
    gen = make_generator(ref_model, 80, '32_1', '3_4_8', [2, 2], 8, 'ERB')
    sd0 = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    mods = []
    for k, v in gen.named_parameters():                      # main_eval.py:296-302
        if 'weight' in k and 'stem' in k:
            mods.append((f'stem.{int(k.split(".")[1])}', gen.stem[int(k.split('.')[1])]))
    for li, layer in enumerate(gen.layers):                  # main_eval.py:305-340
        for b in ERB_BRANCH_MODULES:
            if hasattr(layer, b):
                mods.append((f'layers.{li}.{b}', getattr(layer, b)))
    prune.global_unstructured([(m, 'weight') for _, m in mods], pruning_method=prune.L1Unstructured, amount=0.4)
    out['q1/pruned_modules'] = np.array([n for n, _ in mods])
    for n, m in mods:
        out[f'q1/mask/{n}.weight'] = _np(m.weight_mask)
    fused0 = [layer.get_equivalent_kernel_bias() for layer in gen.layers]
    for li, (wf, bf) in enumerate(fused0):
        out[f'q1/fused0/{li}/wf'] = _np(wf)
        out[f'q1/fused0/{li}/bf'] = _np(bf)
    opt = torch.optim.Adam(gen.parameters(), betas=(0.5, 0.999))
    g = torch.Generator().manual_seed(5)
    frames = torch.rand(3, 3, 12, 16, generator=g)
    pos = torch.tensor([0.0, 1.0 / 3, 2.0 / 3], dtype=torch.float32)
    out['q1/frames'] = _np(frames)
    out['q1/embeds'] = _np(pe(pos))
    lr = 1e-3
    out['q1/lr'] = np.array([lr])
    losses = []
    for it in range(3):
        img = gen(pe(pos[it:it + 1]))[0]
        loss = torch.mean(torch.abs(img - frames[it:it + 1]))            # loss_type L1 (utils.py:143-144)
        for gp in opt.param_groups:
            gp['lr'] = lr
        opt.zero_grad()
        loss.backward(retain_graph=True)                               # main_eval.py:480
        if it == 0:
            opt.state.clear()                                          # main_eval.py:496-497
        opt.step()
        losses.append(float(loss))
    out['q1/losses'] = np.array(losses)
    sd1 = gen.state_dict()
    out['q1/keys_after'] = np.array(list(sd1.keys()))
    changed = []
    for k, v in sd1.items():
        base = sd0.get(k, sd0.get(k.replace('weight_orig', 'weight')))
        if base is not None and not torch.equal(v, base):
            changed.append(k)
        if not k.endswith('_mask'):
            out[f'q1/sd_after/{k}'] = _np(v)
    out['q1/changed'] = np.array(changed)
    for li, layer in enumerate(gen.layers):
        wf, bf = layer.get_equivalent_kernel_bias()
        out[f'q1/fused3/{li}/wf'] = _np(wf)
        out[f'q1/fused3/{li}/bf'] = _np(bf)
    # ---- Q3: deploy-mode ERB, global L1 prune 0.4, then the quantisation walk (main_eval.py:571-587, 659-669) ---------
    torch.manual_seed(1)
    gen = make_generator(ref_model, 80, '32_1', '3_4_8', [2, 2], 8, 'ERB')
    for layer in gen.layers:
        layer.switch_to_deploy()
    dsd0 = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    for k, v in dsd0.items():
        out[f'q3/deploy_sd/{k}'] = _np(v)
    mods = []
    for k, v in gen.named_parameters():
        if 'weight' in k and 'stem' in k:
            mods.append(gen.stem[int(k.split('.')[1])])
    for layer in gen.layers:
        if hasattr(layer, 'rbr_reparam'):
            mods.append(layer.rbr_reparam)
    prune.global_unstructured([(m, 'weight') for m in mods], pruning_method=prune.L1Unstructured, amount=0.4)
    cur = gen.state_dict()
    out['q3/keys'] = np.array(list(cur.keys()))
    levels = []
    for k, v in cur.items():                                 # main_eval.py:660-669
        large_tf = (v.dim() in {2, 4} and 'bias' not in k)
        quant_v, new_v = ref_utils.quantize_per_tensor(v, 8, 0 if large_tf else -1)
        valid = quant_v.detach().cpu()[(v.detach().cpu() != 0)]
        levels.append(valid.flatten())
        out[f'q3/new/{k}'] = _np(new_v)
        out[f'q3/n_valid/{k}'] = np.array([valid.numel()])
    cat = torch.cat(levels)
    uniq, counts = np.unique(np.array(cat.tolist()), return_counts=True)       # main_eval.py:676-677
    out['q3/level_values'] = uniq
    out['q3/level_counts'] = counts
    out['q3/n_symbols'] = np.array([cat.numel()])
    # ---- Q2: the fine-tune LR (main_eval.py:474: adjust_lr(optimizer, epoch % total_epochs, i, data_size, args) with epoch
    # continuing from the checkpoint's 300, total_epochs = 300 + 100, args.epochs = 300, args.warmup = int(0.2 * 300)) -------
    class _A:
        lr, epochs, warmup, lr_type, lr_steps = 5e-4, 300, int(0.2 * 300), 'cosine', []

    class _Opt:
        param_groups = [{'lr': 0.0}]
    tab = []
    for epoch in range(300, 400):
        for it in (0, 66):
            tab.append([epoch, it, ref_utils.adjust_lr(_Opt, epoch % 400, it, 132, _A)])
    out['q2/finetune_lr'] = np.array(tab, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'prune.npz'), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref_model, ref_utils = _import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == 'prune':      # regenerate only the G9 fixture
        golden_prune(ref_model, ref_utils)
        return
    golden_merge(ref_model)
    golden_block(ref_model)
    golden_generator(ref_model, ref_utils)
    golden_utils(ref_utils)
    golden_prune(ref_model, ref_utils)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
