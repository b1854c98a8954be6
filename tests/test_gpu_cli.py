"""End-to-end CLI mirrors (main_train.py / main_eval.py flags of the reference README) on the GPU: short ERB run on a
synthetic clip, checkpoints in the reference's layout, then prune + quantise + evaluate (BASELINE config 5 flow)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

FLAGS = ('-e 4 --lower_width 96 --num_blocks 1 --dataset bunny --frame_gap 1 --embed 1.25_40 --stem_dim_num 512_1 '
         '--reduction 2 --fc_hw_dim 9_16_26 --expansion 1 --single_res --loss Fusion6 --warmup 0.2 --lr_type cosine '
         '--strides 5 2 2 2 2 --conv_type conv -b 1 --lr 0.0005 --norm none --act swish --outf bunny_erb_t --branch_type ERB '
         '--synthetic 12 --eval_freq 2').split()


def test_train_then_eval_cli(tmp_path, monkeypatch):
    import orn_amd
    from orn_amd import checkpoint, main_eval, main_train
    monkeypatch.chdir(tmp_path)
    best = main_train.train(main_train.parse_args(FLAGS))
    assert list(best) == ['synthetic0']
    best = best['synthetic0']
    assert 10.0 < best < 60.0
    outf = tmp_path / 'result' / 'bunny_erb_t'
    # N1 bookkeeping of main_train.py:293-367: latest / train-best (train + deploy) and val-best (train-mode only) files,
    # real best-so-far entries, the "Deploy Rep-Model Params" log line
    for f in ('model_latest.pth', 'model_latest_deploy.pth', 'model_train_best.pth', 'model_train_best_deploy.pth',
              'model_val_best.pth', 'rank0.txt'):
        assert (outf / f).exists(), f
    assert not (outf / 'model_val_best_deploy.pth').exists()
    ck = torch.load(outf / 'model_latest.pth', map_location='cpu', weights_only=True)
    assert set(ck) >= {'epoch', 'state_dict', 'train_best_psnr', 'train_best_msssim', 'val_best_psnr', 'val_best_msssim',
                       'optimizer'} and ck['epoch'] == 4
    assert abs(float(ck['train_best_psnr']) - best) < 1e-4
    assert 0.0 < float(ck['train_best_msssim']) <= 1.0 and 0.0 < float(ck['val_best_msssim']) <= 1.0
    assert float(ck['val_best_psnr']) > 10.0 and float(ck['val_best_psnr']) != float(ck['train_best_psnr'])
    assert float(ck['optimizer']['state'][0]['step']) == 4 * 12
    tb = torch.load(outf / 'model_train_best.pth', map_location='cpu', weights_only=True)
    assert 1 <= tb['epoch'] <= 4 and abs(float(tb['train_best_psnr']) - best) < 1e-4
    if tb['epoch'] != 4:                              # the best epoch's own state, not the last one's
        assert not torch.equal(tb['state_dict']['stem.0.weight'], ck['state_dict']['stem.0.weight'])
    log = (outf / 'rank0.txt').read_text()
    assert 'Deploy Rep-Model Params: 3.202M' in log and 'Eval best_PSNR at epoch4' in log
    sd = checkpoint.load_state_dict_file(str(outf / 'model_latest_deploy.pth'))
    assert checkpoint.state_dict_kind(sd) == 'deploy' and len(sd) == 4 + 5 * 2 + 2
    psnr_plain = main_eval.main(FLAGS)
    psnr_pq = main_eval.main(FLAGS + ['--prune_ratio', '0.4', '--quant_bit', '8', '--dump_images'])
    from PIL import Image
    im = Image.open(outf / 'visualize' / 'pred_11.png')
    assert im.size == (1280, 720) and im.mode == 'RGB'
    assert abs(psnr_plain - best) < 3.0          # decode of the deploy file reproduces the training-time quality
    assert psnr_pq <= psnr_plain + 0.5 and psnr_pq > 5.0
    # BASELINE config 5: prune 0.4 -> fine-tune -> quantise.  Reference quirk Q1: with ERB the branch conv
    # weights stay frozen at their pruned-at-t0 values; stem weights train under their mask; biases train.
    before = checkpoint.load_state_dict_file(str(outf / 'model_latest.pth'))
    captured = {}
    orig = main_eval._prune_finetune

    def spy(model, args, PE, path):
        r = orig(model, args, PE, path)
        captured.update({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
        return r
    monkeypatch.setattr(main_eval, '_prune_finetune', spy)
    psnr_ft = main_eval.main(FLAGS + ['--prune_ratio', '0.4', '--quant_bit', '8', '--finetune', '--finetune_epochs', '3'])
    assert psnr_ft > 5.0
    k = 'layers.3.rbr_1x1_3x3_1x1_branch_3x3.weight'
    pruned = captured[k] == 0
    assert 0.05 < float(pruned.float().mean()) < 0.95
    assert torch.equal(captured[k][~pruned], before[k][~pruned])                 # frozen: survivors bit-unchanged
    ks = 'stem.2.weight'
    zs = captured[ks] == 0
    assert float(zs.float().mean()) > 0.05 and not torch.equal(captured[ks][~zs], before[ks][~zs])   # masked, trained
    kb = 'layers.3.rbr_3x3_branch.bias'
    assert not torch.equal(captured[kb], before[kb])                             # biases train


def test_prune_finetune_quirk_matches_reference_trace():
    """G9 / Q1 against the reference's own trace (tests/golden/prune.npz, made with the reference Generator +
    torch.nn.utils.prune.global_unstructured + 3 Adam steps, main_eval.py:296-350,450-499): same tiny train-mode ERB model,
    same seeded init, global L1 prune 0.4 over the stem Linear weights and all six branch convs of every block.  The masks
    must be identical; after three L1-loss steps on the engine with the fine-tune's gradient masks, the fused kernel of every
    block is bit-identical to the one at prune time (frozen), the bias sum bf moves, and every tensor the reference trains
    -- stem weights under their mask, all biases, the head -- lands on the reference's values (fp32, 1e-6 abs)."""
    import numpy as np
    import orn_amd
    from orn_amd import engine, eval_utils, model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = np.load(os.path.join(root, 'tests', 'golden', 'prune.npz'))
    gg = np.load(os.path.join(root, 'tests', 'golden', 'generator.npz'))
    torch.manual_seed(1)
    gen = model.Generator(embed_length=80, stem_dim_num='32_1', fc_hw_dim='3_4_8', expansion=1, num_blocks=1, norm='none', act='swish',
                          bias=True, reduction=2, conv_type='conv', stride_list=[2, 2], sin_res=True, lower_width=8, sigmoid=False,
                          deploy=False, branch_type='ERB')
    sd0 = {k: torch.from_numpy(gg[f'tiny_ERB/sd/{k}']) for k in (str(x) for x in gg['tiny_ERB/keys'])}
    gen.load_state_dict(sd0)
    named = dict(gen.named_parameters())
    mods = [str(m) for m in g['q1/pruned_modules']]
    prunable = {m + '.weight': named[m + '.weight'].detach() for m in mods}
    masks = eval_utils.global_l1_prune_masks(prunable, 0.4)
    for k, m in masks.items():
        assert torch.equal(m, torch.from_numpy(g[f'q1/mask/{k}'])), k
    with torch.no_grad():
        for k, m in masks.items():
            named[k].mul_(m)
    eng = engine.TrainEngine(gen, loss_type='L1', beta=0.5, precision='fp32')
    eng.set_grad_mask({k: (torch.zeros_like(m) if k.startswith('layers.') else m) for k, m in masks.items()})     # Q1
    eng.set_video(torch.from_numpy(g['q1/frames']), torch.from_numpy(g['q1/embeds']))
    fused0 = [tuple(t.clone() for t in eng.fused_kernel(li)) for li in range(2)]
    for li in range(2):                                                                # same fused kernel as the reference at prune time
        np.testing.assert_allclose(fused0[li][0].cpu().numpy(), g[f'q1/fused0/{li}/wf'], rtol=0, atol=2e-7)
    lr = float(g['q1/lr'][0])
    eng.set_schedule([(0, 1, lr), (1, 2, lr), (2, 3, lr)])
    eng.run(3, graph=False)
    torch.cuda.synchronize()
    st = eng.stats(3)
    np.testing.assert_allclose(st[:, 0].numpy(), g['q1/losses'], rtol=2e-5)
    after = {k: v.detach().cpu() for k, v in gen.state_dict().items()}
    for li in range(2):
        wf, bf = eng.fused_kernel(li)
        assert torch.equal(wf, fused0[li][0]), li                                       # frozen, bit for bit
        assert not torch.equal(bf, fused0[li][1])
        np.testing.assert_allclose(bf.cpu().numpy(), g[f'q1/fused3/{li}/bf'], rtol=0, atol=2e-6)
    changed = {str(k) for k in g['q1/changed']}
    for k in (str(x) for x in g['q1/keys_after']):
        if k.endswith('_mask'):
            continue
        ref = torch.from_numpy(g[f'q1/sd_after/{k}'])
        if k.endswith('weight_orig'):
            mine_k = k[:-5]
            if mine_k.startswith('layers.'):
                # the reference's Adam moves weight_orig, which nothing ever reads again: the engine keeps the pruned values
                assert torch.equal(after[mine_k], sd0[mine_k] * masks[mine_k]), k
                continue
            m = masks[mine_k]
            np.testing.assert_allclose((after[mine_k] * m).numpy(), (ref * m).numpy(), rtol=0, atol=1e-6, err_msg=k)
            assert torch.equal(after[mine_k][m == 0], torch.zeros_like(after[mine_k][m == 0]))
        else:
            assert (k in changed) == (not torch.equal(after[k], sd0[k])), k
            np.testing.assert_allclose(after[k].numpy(), ref.numpy(), rtol=0, atol=1e-6, err_msg=k)


def test_png_directory_through_the_cli(tmp_path, monkeypatch):
    """N4 on the GPU: a PNG directory (7 frames, frame_gap 2 -> 3 training samples at times 0, 2/7, 4/7; test_gap 3 -> validation
    samples 0 and 3) goes through main_train exactly as CustomDataSet would feed it (model.py:11-70): resident frames and the
    embedding table of the engine are checked against the files."""
    import numpy as np
    from PIL import Image
    import orn_amd
    from orn_amd import main_train, utils
    run = tmp_path / 'run'
    run.mkdir()
    d = tmp_path / 'data' / 'tinyvid'
    d.mkdir(parents=True)
    rng = np.random.RandomState(3)
    imgs = []
    for k in range(7):
        a = rng.randint(0, 256, size=(40, 60, 3), dtype=np.uint8)
        Image.fromarray(a).save(d / f'{k:05d}.png')
        imgs.append(a)
    monkeypatch.chdir(run)
    flags = ('-e 2 --lower_width 96 --num_blocks 1 --dataset tinyvid --frame_gap 2 --test_gap 3 --embed 1.25_40 --stem_dim_num 32_1 '
             '--reduction 2 --fc_hw_dim 2_3_26 --expansion 1 --single_res --loss Fusion6 --warmup 0.2 --lr_type cosine --strides 5 2 2 '
             '--conv_type conv -b 1 --lr 0.0005 --norm none --act swish --outf tiny_t --branch_type ERB --eval_freq 1 --precision fp32').split()
    captured = {}
    orig = main_train.evaluate

    def spy(model, eng, args, val=None, gap=None):
        if gap is None:                               # (gap=1: the train MS-SSIM pass over the training frames)
            captured['frames'], captured['embeds'] = eng.frames.clone(), eng.embeds.clone()
            captured['val'] = None if val is None else (val[0].clone(), val[1].clone())
        return orig(model, eng, args, val, gap)
    monkeypatch.setattr(main_train, 'evaluate', spy)
    best = main_train.train(main_train.parse_args(flags))['tinyvid']
    assert best > 5.0
    fr = captured['frames'].cpu()
    assert fr.shape == (3, 3, 40, 60)
    for k in range(3):
        assert torch.equal(fr[k], torch.from_numpy(imgs[2 * k]).permute(2, 0, 1).float() / 255.0)
    PE = utils.PositionalEncoding('1.25_40')
    want = PE(torch.tensor([0.0, 2.0 / 7, 4.0 / 7], dtype=torch.float32)).cpu()
    assert torch.equal(captured['embeds'].cpu(), want)
    vf, ve = captured['val']
    assert vf.shape[0] == 2 and torch.equal(vf[1].cpu(), torch.from_numpy(imgs[3]).permute(2, 0, 1).float() / 255.0)
    assert torch.equal(ve.cpu(), PE(torch.tensor([0.0, 3.0 / 7], dtype=torch.float32)).cpu())
    assert (run / 'result' / 'tiny_t' / 'model_latest.pth').exists()
