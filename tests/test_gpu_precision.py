"""GPU tests of what makes the 16-bit headline mode creditable against an fp32 reference (main_train.py:193-250 trains in
fp32, no autocast): the non-finite guard + dynamic loss scale, and PSNR parity of a whole fit on content that discriminates."""
import json
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def orn():
    import orn_amd
    from orn_amd import ops, model, utils, engine  # noqa: F401
    orn_amd._lib.lib()
    return orn_amd


def _small_engine(orn, prec, n_frames=4):
    from oracle import cpu_ref
    torch.manual_seed(1)
    gen = orn.model.Generator(embed_length=80, stem_dim_num='32_1', fc_hw_dim='2_3_26', expansion=1, num_blocks=1, norm='none',
                              act='swish', bias=True, reduction=2, conv_type='conv', stride_list=[5, 2, 2], sin_res=True,
                              lower_width=96, sigmoid=False, deploy=False, branch_type='ERB')
    eng = orn.engine.TrainEngine(gen, loss_type='Fusion6', beta=0.5, precision=prec)
    hw = eng.out_hw
    frames = cpu_ref.synthetic_video(n_frames, hw[0], hw[1], seed=5)
    embeds = cpu_ref.positional_encoding(torch.tensor([k / n_frames for k in range(n_frames)]), 1.25, 40)
    eng.set_video(frames, embeds)
    return eng


def test_overflowing_steps_are_skipped_and_the_scale_backs_off(orn):
    """fp16 mode with the gradient scale forced to 2^40: |dL/dz| * 2^40 overflows IEEE half, so the 16-bit gradient tensors
    carry inf.  The engine must (1) detect it on the device (no host sync inside the captured step), (2) leave parameters
    AND Adam moments bit-identical for every such step, (3) halve the scale at each schedule advance until the gradients fit,
    (4) then train normally: finite loss, parameters moving, step count excluding the skipped steps."""
    eng = _small_engine(orn, 'fp16')
    eng.set_schedule([(k % 4, k + 1, 5e-4) for k in range(200)])
    eng.run(4, graph=True)                                    # healthy steps first
    torch.cuda.synchronize()
    s0 = eng.scale_state()
    assert s0['skipped'] == 0 and s0['scale'] == 2.0 ** 20 and s0['flag'] == 0
    eng.set_grad_scale(2.0 ** 40)
    p0, m0, v0 = eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone()
    eng.run(4, graph=True)                                    # one unrolled graph = one advance: all four steps overflow
    torch.cuda.synchronize()
    s1 = eng.scale_state()
    assert s1['skipped'] == 4 and s1['flag'] == 1, s1
    assert torch.equal(eng.params, p0) and torch.equal(eng.adam_m, m0) and torch.equal(eng.adam_v, v0)
    eng.run(120, graph=True)                                  # every advance halves the scale until the step fits
    torch.cuda.synchronize()
    s2 = eng.scale_state()
    assert s2['backoffs'] >= 10 and s2['scale'] < 2.0 ** 31 and s2['scale'] >= 1.0, s2
    assert s2['skipped'] < 124, s2                            # it did start stepping again
    assert torch.isfinite(eng.params).all() and torch.isfinite(eng.adam_m).all() and torch.isfinite(eng.adam_v).all()
    assert not torch.equal(eng.params, p0)
    st = eng.stats(128)
    assert torch.isfinite(st[:, 0]).all()
    # Adam's step count (stats column 7) excludes the skipped steps, as a skipped optimizer.step() does in torch
    assert int(st[127, 7]) <= 128 - 4


@pytest.mark.parametrize('prec', ['fp32', 'bf16', 'fp16'])
def test_a_nan_frame_poisons_nothing(orn, prec):
    """A frame containing NaN makes loss and every gradient NaN.  In every precision mode the step must be skipped (loss
    guard in the loss kernel), parameters and moments stay finite and unchanged, and training on the clean frames goes on."""
    eng = _small_engine(orn, prec)
    eng.frames[2, 1, 3, 5] = float('nan')
    eng.set_schedule([(0, 1, 5e-4), (1, 2, 5e-4), (3, 3, 5e-4), (0, 4, 5e-4), (2, 5, 5e-4), (2, 6, 5e-4), (2, 7, 5e-4), (2, 8, 5e-4),
                      (0, 9, 5e-4), (1, 10, 5e-4), (3, 11, 5e-4), (0, 12, 5e-4)])
    eng.run(4, graph=True)
    torch.cuda.synchronize()
    p0, m0 = eng.params.clone(), eng.adam_m.clone()
    eng.run(4, graph=True)                                    # four steps on the poisoned frame
    torch.cuda.synchronize()
    assert torch.equal(eng.params, p0) and torch.equal(eng.adam_m, m0)
    assert eng.scale_state()['skipped'] == 4
    eng.run(4, graph=True)
    torch.cuda.synchronize()
    assert torch.isfinite(eng.params).all() and not torch.equal(eng.params, p0)
    assert torch.isfinite(eng.stats(12)[8:, 0]).all()

@pytest.mark.parametrize('prec', ['fp32', 'fp16'])
def test_one_bad_step_skips_only_itself(orn, prec):
    """The non-finite flag is per step of the unrolled graph: ONE poisoned step inside a group of four skips itself only (the clean
    steps before and behind it in the same graph launch update the parameters), is counted once, and backs the scale off once."""
    eng = _small_engine(orn, prec)
    eng.frames[2, 1, 3, 5] = float('nan')
    eng.set_schedule([(0, 1, 5e-4), (1, 2, 5e-4), (3, 3, 5e-4), (0, 4, 5e-4),        # clean group
                      (0, 5, 5e-4), (2, 6, 5e-4), (1, 7, 5e-4), (3, 8, 5e-4),        # second step poisoned
                      (0, 9, 5e-4), (1, 10, 5e-4), (3, 11, 5e-4), (0, 12, 5e-4)])
    eng.run(4, graph=True)
    torch.cuda.synchronize()
    ref = _small_engine(orn, prec)                              # the same fit without the poisoned step
    ref.params.copy_(eng.params); ref.adam_m.copy_(eng.adam_m); ref.adam_v.copy_(eng.adam_v)
    eng.run(4, graph=True)
    torch.cuda.synchronize()
    s = eng.scale_state()
    assert s['skipped'] == 1 and s['flag'] == 1, s
    st = eng.stats(8)
    assert torch.isfinite(st[[4, 6, 7], 0]).all() and not torch.isfinite(st[5, 0])
    # Adam step numbers in the device schedule are those of the launch (5, 6, 7, 8 minus the skips known at the advance: none)
    ref.set_schedule([(0, 5, 5e-4), (1, 7, 5e-4), (3, 8, 5e-4)])
    ref.run(1, graph=False); ref.run(1, graph=False); ref.run(1, graph=False)
    torch.cuda.synchronize()
    assert torch.isfinite(eng.params).all()
    assert torch.allclose(eng.params, ref.params, rtol=0, atol=1e-6 if prec == 'fp32' else 1e-4)
    eng.run(4, graph=True)
    torch.cuda.synchronize()
    s2 = eng.scale_state()
    assert s2['skipped'] == 1 and s2['backoffs'] == 1 and s2['flag'] == 0, s2


def test_fp16_fit_matches_fp32_fit_psnr(orn):
    """PSNR parity of a whole fit, fp16 engine vs fp32 engine (the reference's arithmetic), north_star tolerance 0.05 dB:
    BASELINE config 2 (720p, ERB, 9_16_26) with the reference recipe (lr 5e-4, warm-up 0.2, cosine to zero, Adam(0.5, 0.999),
    b = 1, shuffled epochs), 60 epochs over a 48-frame video of drifting band-limited textures -- no white noise, so there is
    no noise floor to hide behind; the fit ends at ~37 dB, above the 35 dB where the comparison starts to discriminate and
    below the >50 dB regime where single ulps decide.  Three seeds (video content AND model initialisation).  Asserted: every
    fit passes 35 dB, mean |dPSNR| <= 0.05 dB, no step skipped.  For scale: two fp32 fits that differ by ONE ulp in one initial
    weight end 0.018 dB apart on this content (profiles/r02_psnr_parity_texture14_60ep_*.jsonl) -- the trajectory is chaotic
    at that level, so this is also the resolution of the comparison.  The per-seed table goes to gpurun_out/."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import psnr_parity
    import bench
    rows, deltas = [], []
    try:
        _fits(psnr_parity, rows, deltas)
    finally:
        bench.CFG.update(frames=132, epochs=300, warmup=60)           # psnr_parity.run changes these module globals
    out = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'psnr_parity_test.jsonl'), 'w') as f:
            for r in rows:
                f.write(json.dumps(r) + '\n')
            f.write(json.dumps({'fp16_minus_fp32_dB': deltas, 'mean_abs_dB': sum(abs(d) for d in deltas) / 3}) + '\n')
    assert sum(abs(d) for d in deltas) / 3 <= 0.05, deltas


def _fits(psnr_parity, rows, deltas):
    for seed in range(3):
        r32 = psnr_parity.run('fp32', 60, 48, seed, 0.0, 'texture', 14.0)
        r16 = psnr_parity.run('fp16', 60, 48, seed, 0.0, 'texture', 14.0)
        rows += [r32, r16]
        deltas.append(r16['eval_psnr'] - r32['eval_psnr'])
        assert r32['eval_psnr'] > 35.0 and r16['eval_psnr'] > 35.0, (r32['eval_psnr'], r16['eval_psnr'])
        assert r16['skipped'] == 0 and r16['scale'] == 2.0 ** 20


def test_dead_fp16_fit_falls_back_to_a_wider_precision(orn, tmp_path, monkeypatch):
    """A fit that leaves fp16's range (forward activations overflow, every later step is skipped, the run
    burns its remaining epochs) is detected at the end of the epoch, taken back to the start of that epoch -- parameters, Adam
    moments, step count -- and continued in bf16: no restart.  Forced here by blowing up one branch kernel after the epoch's
    snapshot was taken (1e9 x: the merged kernel itself overflows IEEE half); the restore undoes the damage, so the bf16 continuation
    trains normally."""
    from orn_amd import main_train
    monkeypatch.chdir(tmp_path)
    flags = ('-e 4 --lower_width 96 --num_blocks 1 --dataset x --frame_gap 1 --embed 1.25_40 --stem_dim_num 32_1 --reduction 2 '
             '--fc_hw_dim 2_3_26 --expansion 1 --single_res --loss Fusion6 --warmup 0.2 --lr_type cosine --strides 5 2 2 --conv_type conv '
             '-b 1 --lr 0.0005 --norm none --act swish --outf dead --branch_type ERB --synthetic 6 --eval_freq 100 --precision fp16').split()
    args = main_train.parse_args(flags)
    seen = {}

    def inject(epoch, eng):
        seen.setdefault('prec', []).append(eng.precision)
        if epoch == 1 and 'hit' not in seen:
            seen['hit'] = True
            off, n = eng.layout['layers.2.rbr_3x3_branch.weight']
            eng.params[off:off + n] *= 1.0e9
    best, frames, steps = main_train.fit_video(args, 'synthetic0', 0, 0, _inject=inject)
    # epochs seen by the hook: 0 (fp16), 1 (fp16, poisoned), 1 again (bf16, restored), 2, 3 (bf16)
    assert seen['prec'] == [2, 2, 1, 1, 1], seen
    assert math.isfinite(best) and best > 5.0
    log = (tmp_path / 'result' / 'dead' / 'rank0.txt').read_text()
    assert 'Restoring the start of epoch 2 and continuing in --precision bf16' in log
    assert 'precision: started in fp16, finished in bf16' in log
    ck = torch.load(tmp_path / 'result' / 'dead' / 'model_latest.pth', map_location='cpu', weights_only=True)
    assert all(torch.isfinite(v).all() for v in ck['state_dict'].values())
    assert float(ck['state_dict']['layers.2.rbr_3x3_branch.weight'].abs().max()) < 10.0       # the blow-up is gone


@pytest.mark.parametrize('prec', ['fp32', 'fp16'])
def test_target_statistics_cache_changes_nothing(orn, prec):
    """Fusion6's target-side SSIM statistics (G*t, G*t^2) read from the per-video table (orn_loss_target_stats) instead of being
    filtered in every step: same taps, same fmaf order -- losses, PSNR and every parameter after 12 steps are bit-identical."""
    from oracle import cpu_ref
    res = []
    for cache in (False, True):
        torch.manual_seed(1)
        gen = orn.model.Generator(embed_length=80, stem_dim_num='32_1', fc_hw_dim='2_3_26', expansion=1, num_blocks=1, norm='none',
                                  act='swish', bias=True, reduction=2, conv_type='conv', stride_list=[5, 2, 2], sin_res=True,
                                  lower_width=96, sigmoid=False, deploy=False, branch_type='ERB')
        eng = orn.engine.TrainEngine(gen, loss_type='Fusion6', beta=0.5, precision=prec, target_cache=cache)
        frames = cpu_ref.synthetic_video(5, eng.out_hw[0], eng.out_hw[1], seed=5)
        embeds = cpu_ref.positional_encoding(torch.tensor([k / 5 for k in range(5)]), 1.25, 40)
        eng.set_video(frames, embeds)
        assert (eng.tstats is not None) == cache
        eng.set_schedule([(k % 5, k + 1, 5e-4) for k in range(12)])
        eng.run(12, graph=True)
        torch.cuda.synchronize()
        res.append((eng.stats(12).clone(), eng.params.clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
