"""GPU tests of the pipelined form of the step (orn_engine_train_steps, include/orn.h): the last block's weight gradient, slab
reduction, merge backward, Adam update and next merge forward run on the engine's second stream, beside the boundary between this
step and the next (main_train.py:229-254 is still the unit of work).  It must change nothing but the clock: parameters, Adam moments
and the per-step statistics are compared BIT FOR BIT with the serial forms of the step (hipGraph replay, one call per step), whose
gradients the other test files pin to the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def orn():
    import orn_amd
    from orn_amd import ops, model, utils, engine  # noqa: F401
    orn_amd._lib.lib()
    return orn_amd


GEOS = {
    # two blocks with 96 input channels: both on the 16-bit path, no fp32 block below them
    'c96x2': dict(fc='3_4_96', strides=[2, 2], lower_width=96),
    # the bench geometry in small: an fp32 first block (26 channels), a narrow block, two 96-channel blocks
    'narrow_first': dict(fc='2_3_26', strides=[5, 2, 2, 2], lower_width=96),
    # a stride-3 block in the middle (config 3's shape)
    'stride3': dict(fc='2_3_26', strides=[5, 3, 2], lower_width=96),
}


def _engine(orn, prec, branch, geo, n_frames=5, seed=1):
    from oracle import cpu_ref
    g = GEOS[geo]
    torch.manual_seed(seed)
    gen = orn.model.Generator(embed_length=80, stem_dim_num='32_1', fc_hw_dim=g['fc'], expansion=1, num_blocks=1, norm='none',
                              act='swish', bias=True, reduction=2, conv_type='conv', stride_list=g['strides'], sin_res=True,
                              lower_width=g['lower_width'], sigmoid=False, deploy=False, branch_type=branch)
    eng = orn.engine.TrainEngine(gen, loss_type='Fusion6', beta=0.5, precision=prec)
    hw = eng.out_hw
    frames = cpu_ref.synthetic_video(n_frames, hw[0], hw[1], seed=5)
    embeds = cpu_ref.positional_encoding(torch.tensor([k / n_frames for k in range(n_frames)]), 1.25, 40)
    eng.set_video(frames, embeds)
    return eng


def _run(orn, prec, branch, geo, mode, steps, calls):
    eng = _engine(orn, prec, branch, geo)
    eng.set_schedule([(k % 5, k + 1, 5e-4) for k in range(steps * calls)])
    for _ in range(calls):
        eng.run(steps, graph=mode)
    torch.cuda.synchronize()
    return eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng.stats(steps * calls).clone()


@pytest.mark.parametrize('geo', sorted(GEOS))
@pytest.mark.parametrize('branch', ['ERB', 'NeRV_vanilla'])
@pytest.mark.parametrize('prec', ['fp16', 'bf16'])
def test_pipelined_steps_equal_serial_steps_bit_for_bit(orn, prec, branch, geo):
    """7 steps as two calls (4 + 3: the first step of a call merges every block itself, the later ones take the last block's
    merged kernel from the previous step's side branch; the end of a call joins the branch)."""
    ref = _run(orn, prec, branch, geo, True, 7, 1)            # hipGraph replay of the serial step
    pipe_a = _run(orn, prec, branch, geo, None, 7, 1)
    for a, b, what in zip(ref, pipe_a, ('params', 'adam_m', 'adam_v', 'stats')):
        assert torch.equal(a, b), (what, float((a - b).abs().max()))
    eng = _engine(orn, prec, branch, geo)
    eng.set_schedule([(k % 5, k + 1, 5e-4) for k in range(7)])
    eng.run(4)
    eng.run(3)
    torch.cuda.synchronize()
    assert torch.equal(eng.params, ref[0]) and torch.equal(eng.adam_m, ref[1]) and torch.equal(eng.adam_v, ref[2])
    assert torch.isfinite(ref[0]).all() and not torch.equal(ref[1], torch.zeros_like(ref[1]))


@pytest.mark.parametrize('cfg', ['720p', '1080p'])
def test_pipelined_steps_equal_eager_steps_at_720p(orn, cfg):
    """BASELINE config 2 (720p, 9_16_26) and config 3's geometry (1080p, 9_16_48, a stride-3 block) at full size: 6 pipelined steps
    against 6 single-step calls, bit for bit."""
    import bench
    outs = []
    for mode in (False, None):
        eng = bench.make_engine(seed=7, precision='fp16', cfg=bench.CONFIGS[cfg], frames=6)
        eng.set_schedule([(k % 6, k + 1, 5e-4) for k in range(6)])
        eng.run(6, graph=mode)
        torch.cuda.synchronize()
        outs.append((eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng.stats(6).clone()))
        del eng
    for a, b, what in zip(outs[0], outs[1], ('params', 'adam_m', 'adam_v', 'stats')):
        assert torch.equal(a, b), (what, float((a - b).abs().max()))


def test_pipelined_epochs_equal_graph_replay_at_720p(orn):
    """A soak for the hand-offs between the streams (a missed dependency shows as a different bit sooner or later): 3 epochs of the
    132-frame bench video, shuffled schedule with the reference's LR ramp, as pipelined calls of one epoch each against the hipGraph
    replay of the serial step -- parameters, both moments and all 396 per-step records bit for bit -- and the pipelined run repeated
    (run-to-run identical)."""
    import bench
    outs = []
    for mode in (True, None, None):
        eng = bench.make_engine(seed=11, precision='fp16', cfg=bench.CONFIGS['720p'])
        eng.set_schedule(bench.schedule(396))
        for _ in range(3):
            eng.run(132, graph=mode)
        torch.cuda.synchronize()
        outs.append((eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng.stats(396).clone()))
        assert eng.scale_state()['skipped'] == 0
        del eng
    for k in (1, 2):
        for a, b, what in zip(outs[0], outs[k], ('params', 'adam_m', 'adam_v', 'stats')):
            assert torch.equal(a, b), (k, what, float((a - b).abs().max()))


def test_a_skipped_step_is_skipped_on_both_streams(orn):
    """The guard under the pipeline: a step whose gradients overflow (scale forced to 2^40) must leave EVERY parameter and moment
    bit-identical -- the side branch's Adam launch (last block + head) follows the decision the main stream took -- and count once."""
    eng = _engine(orn, 'fp16', 'ERB', 'narrow_first')
    eng.set_schedule([(k % 5, k + 1, 5e-4) for k in range(40)])
    eng.run(3)
    torch.cuda.synchronize()
    assert eng.scale_state()['skipped'] == 0
    eng.set_grad_scale(2.0 ** 40)
    p0, m0, v0 = eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone()
    eng.run(3)
    torch.cuda.synchronize()
    s = eng.scale_state()
    assert s['skipped'] == 3, s
    assert torch.equal(eng.params, p0) and torch.equal(eng.adam_m, m0) and torch.equal(eng.adam_v, v0)
    eng.run(30)                                                # every step advances: the scale halves until the gradients fit
    torch.cuda.synchronize()
    s = eng.scale_state()
    assert s['skipped'] < 33 and torch.isfinite(eng.params).all() and not torch.equal(eng.params, p0), s


def test_decode_after_pipelined_steps_uses_the_updated_last_block(orn):
    """The end of a call joins the side branch: a decode right behind it must see the last block's and the head's updated
    parameters (same image as after the same steps run serially)."""
    imgs = []
    for mode in (True, None):
        eng = _engine(orn, 'fp16', 'ERB', 'narrow_first')
        eng.set_schedule([(k % 5, k + 1, 5e-4) for k in range(5)])
        eng.run(5, graph=mode)
        imgs.append(eng.decode(eng.embeds[2]).clone())
        torch.cuda.synchronize()
    assert torch.equal(imgs[0], imgs[1])
