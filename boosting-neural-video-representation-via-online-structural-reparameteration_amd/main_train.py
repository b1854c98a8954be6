"""`python -m orn_amd.main_train <flags>`: the reference's training CLI (main_train.py:38-157) driving
the native engine.  Same flags; the data source is a synthetic video resident in HBM (`--synthetic N`)
or a directory of PNG frames (`../data/<dataset>` as in main_train.py:202-203, needs PIL).

Differences kept deliberately small and stated here: no tensorboard/thop (not in this image); checkpoint
FILES are written every `--ckpt_freq` epochs (default: every eval epoch and the last) instead of every
epoch, because an epoch takes a fraction of a second here -- the state of the best-train-PSNR
epoch is kept as a snapshot in HBM the moment it happens (three 30 MB device copies), so
`model_train_best(.pth|_deploy.pth)` holds exactly the epoch the reference would have saved; per-rank
videos under torch.distributed.run (one independent fit per GPU, RCCL only for the final gather).
The reference's arithmetic is `--precision fp32`; the default here is fp16 (16-bit MFMA, fp32 master
weights, dynamic loss scale), and a fit that leaves the 16-bit range falls back to the next wider
precision from the start of the offending epoch (see `fit_video`).
"""
import argparse
import os
import sys
import time

import torch

from . import data as odata
from . import dist_utils as du
from . import engine as oeng
from . import model as omodel
from . import ops, utils


def build_parser():
    p = argparse.ArgumentParser(fromfile_prefix_chars='@')
    p.add_argument('--vid', default=[None], type=int, nargs='+')
    p.add_argument('--scale', type=int, default=1)
    p.add_argument('--frame_gap', type=int, default=1)
    p.add_argument('--augment', type=int, default=0)
    p.add_argument('--dataset', type=str, default='UVG')
    p.add_argument('--test_gap', default=1, type=int)
    p.add_argument('--embed', type=str, default='1.25_80')
    p.add_argument('--stem_dim_num', type=str, default='1024_1')
    p.add_argument('--fc_hw_dim', type=str, default='9_16_128')
    p.add_argument('--expansion', type=float, default=8)
    p.add_argument('--reduction', type=int, default=2)
    p.add_argument('--strides', type=int, nargs='+', default=[5, 3, 2, 2, 2])
    p.add_argument('--num_blocks', type=int, default=1)
    p.add_argument('--norm', default='none', type=str, choices=['none', 'bn', 'in'])
    p.add_argument('--act', type=str, default='gelu',
                   choices=['relu', 'leaky', 'leaky01', 'relu6', 'gelu', 'swish', 'softplus', 'hardswish'])
    p.add_argument('--lower_width', type=int, default=32)
    p.add_argument('--single_res', action='store_true')
    p.add_argument('--conv_type', default='conv', type=str, choices=['conv', 'deconv', 'bilinear'])
    p.add_argument('--branch_type', default='NeRV_vanilla', type=str,
                   choices=['NeRV_vanilla', 'ERB', 'ACB', 'RepVGG', 'DBB', 'ECB'])
    p.add_argument('-j', '--workers', type=int, default=4)
    p.add_argument('-b', '--batchSize', type=int, default=1)
    p.add_argument('--not_resume_epoch', action='store_true')
    p.add_argument('-e', '--epochs', type=int, default=150)
    p.add_argument('--warmup', type=float, default=0.2)
    p.add_argument('--lr', type=float, default=0.001)
    p.add_argument('--lr_type', type=str, default='cosine')
    p.add_argument('--lr_steps', default=[], type=float, nargs='+')
    p.add_argument('--beta', type=float, default=0.5)
    p.add_argument('--loss_type', type=str, default='L2')
    p.add_argument('--lw', type=float, default=1.0)
    p.add_argument('--sigmoid', action='store_true')
    p.add_argument('--deploy', action='store_true', default=False)
    p.add_argument('--eval_only', action='store_true', default=False)
    p.add_argument('--eval_freq', type=int, default=50)
    p.add_argument('--quant_bit', type=int, default=-1)
    p.add_argument('--quant_axis', type=int, default=0)
    p.add_argument('--dump_images', action='store_true', default=False)
    p.add_argument('--eval_fps', action='store_true', default=False)
    p.add_argument('--prune_steps', type=float, nargs='+', default=[0., ])
    p.add_argument('--prune_ratio', type=float, default=1.0)
    p.add_argument('--manualSeed', type=int, default=1)
    p.add_argument('--init_method', default='tcp://127.0.0.1:9888', type=str)
    p.add_argument('-d', '--distributed', action='store_true', default=False)
    p.add_argument('--debug', action='store_true')
    p.add_argument('-p', '--print_freq', default=50, type=int)
    p.add_argument('--weight', default='None', type=str)
    p.add_argument('--overwrite', action='store_true')
    p.add_argument('--outf', default='unify')
    p.add_argument('--suffix', default='')
    # additions of this engine
    p.add_argument('--precision', default='fp16', choices=['fp32', 'bf16', 'fp16'],
                   help='conv arithmetic: fp32 is the reference\'s own (main_train.py has no autocast); fp16 / bf16 run the convs '
                        'on 16-bit MFMA with fp32 accumulation and fp32 master weights (default fp16: ~8x faster, PSNR within '
                        '0.05 dB; falls back to bf16, then fp32, if a fit leaves its range)')
    p.add_argument('--synthetic', type=int, default=0, help='use N synthetic frames in HBM instead of ../data/<dataset>')
    p.add_argument('--synthetic_videos', type=int, default=0,
                   help='with --synthetic: independent synthetic videos of the job (seeds 1234 + v), dealt round-robin to the ranks '
                        '(default: one per rank)')
    p.add_argument('--ckpt_freq', type=int, default=0, help='checkpoint every K epochs (0: eval epochs and the last)')
    p.add_argument('--dist_backend', default=None, choices=['nccl', 'gloo'],
                   help='torch.distributed backend under a multi-process launcher (default: nccl = RCCL on GPUs)')
    return p


def parse_args(argv=None):
    args = build_parser().parse_args(argv)
    args.warmup = int(args.warmup * args.epochs)                   # main_train.py:111
    if args.debug:
        args.eval_freq = 1
        args.outf = 'result/debug'
    else:
        args.outf = os.path.join('result', args.outf)
    args.outf = os.path.join(args.outf, f'{args.suffix}')          # main_train.py:138
    return args


def video_list(args, world):
    """The independent fits of this job (one video per rank, round-robin when there are more): `--dataset a,b,c`
    names one frame directory each (../data/<name>, main_train.py:181); with --synthetic there are max(world, 1) seeded
    synthetic videos."""
    if args.synthetic:
        n = getattr(args, 'synthetic_videos', 0)
        return [f'synthetic{v}' for v in range(n if n > 0 else max(world, 1))]
    return [d for d in args.dataset.split(',') if d]


def video_outf(base_outf, videos, v):
    """Output directory of video v: the job's own directory when it fits a single video (the reference layout), else one
    sub-directory per video -- ranks fit different videos, so no two ranks ever write the same checkpoint or log."""
    return base_outf if len(videos) == 1 else os.path.join(base_outf, videos[v])


def load_frames(args, hw, device, name, vid_index, gap):
    """(frames, times) of one video: CustomDataSet indexing for a frame directory (model.py:11-70), k / n for synthetic."""
    if args.synthetic:
        frames = odata.synthetic_video(args.synthetic, hw[0], hw[1], seed=1234 + vid_index, device=device)
        n = frames.shape[0]
        frames, pos = frames, torch.tensor([float(k) / n for k in range(n)], dtype=torch.float32)
        if gap > 1:
            keep = [k * gap for k in range(n // gap)]
            frames, pos = frames[keep].contiguous(), pos[keep]
        return frames, pos
    return odata.load_png_dir(f'../data/{name.lower()}', args.vid, gap, device)


class Best:
    """The reference's four best-so-far entries of a checkpoint (main_train.py:219,271-273,310-312)."""
    def __init__(self):
        self.train_psnr = torch.tensor(0.0)
        self.train_msssim = torch.tensor(0.0)
        self.val_psnr = torch.tensor(0.0)
        self.val_msssim = torch.tensor(0.0)


def save_checkpoint(args, model, eng, epoch, best, name='model_latest.pth', deploy_too=None, applied=None):
    """main_train.py:293-301,327 layout; ERB also writes the deploy copy (main_train.py:325-351).  `best`: a Best, or a
    number (train PSNR; kept for callers that only track that)."""
    from . import checkpoint
    if not isinstance(best, Best):
        b = Best()
        b.train_psnr = b.val_psnr = torch.as_tensor(float(best))
        best = b
    opt = adam_state_dict(model, eng, args, applied)
    kw = dict(train_best_psnr=best.train_psnr, val_best_psnr=best.val_psnr, train_best_msssim=best.train_msssim,
              val_best_msssim=best.val_msssim)
    checkpoint.save(os.path.join(args.outf, name), model, epoch + 1, opt, **kw)
    if args.branch_type == 'ERB' if deploy_too is None else deploy_too:
        checkpoint.save(os.path.join(args.outf, name.replace('.pth', '_deploy.pth')), model, epoch + 1, opt, deploy=True, **kw)


class Snapshot:
    """Parameters + Adam moments + step count of an engine at one moment, held in HBM (three arena-sized device copies:
    ~40 us at 720p).  Two uses: the state of the best-train-PSNR epoch until the next checkpoint write (the reference
    writes model_train_best.pth in that epoch, main_train.py:329,348,356), and the start of the current epoch for the
    fall-back of a fit that has left its precision's range."""
    def __init__(self, eng, epoch):
        self.params, self.m, self.v = eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone()
        self.step, self.epoch = eng.global_step, epoch
        self.applied = eng.applied_steps()          # optimiser steps actually applied up to this moment (skipped ones excluded)

    def restore(self, eng):
        eng.params.copy_(self.params); eng.adam_m.copy_(self.m); eng.adam_v.copy_(self.v)
        eng.global_step = self.step

    def write(self, args, model, eng, best, name):
        """Write the snapshot as a checkpoint: the engine's arenas are swapped to the snapshot for the duration of the write."""
        now = Snapshot(eng, -1)
        self.restore(eng)
        try:
            save_checkpoint(args, model, eng, self.epoch, best, name, applied=self.applied)     # Adam's count as it stood at the snapshot
        finally:
            now.restore(eng)


def skipped_steps_warning(skipped_before: int, skipped_now: int, steps: int, precision: str, scale: float):
    """The non-finite guard leaves parameters and Adam state untouched by a step whose loss or gradients are not finite and
    backs the loss scale off; a few such steps are its normal work.  An epoch in which MOST steps were skipped means the fit
    itself has left the number range (on some content the reference recipe blows up while its LR is still ramping --
    DESIGN.md section 5 -- and fp16 activations then overflow in the forward, where no scale can help): say so instead of
    training on in silence.  Returns the warning line or None."""
    d = skipped_now - skipped_before
    if steps <= 0 or 2 * d <= steps:
        return None
    return (f'WARNING: {d} of {steps} steps of this epoch were skipped (non-finite loss or gradients; loss scale now {scale:g}). '
            f'The fit has left the range of --precision {precision}; skipped steps change nothing, so it will not recover by itself. '
            f'Re-run with --precision fp32' + (' or bf16' if precision == 'fp16' else '') + ' or a lower --lr.')


def adam_state_dict(model, eng, args, applied=None):
    """The optimizer entry of a checkpoint in torch.optim.Adam's own state_dict layout (per-parameter 'step' / 'exp_avg' /
    'exp_avg_sq' in model.parameters() order + one param group), so the reference's
    `optimizer.load_state_dict(checkpoint['optimizer'])` (main_eval.py:409, main_train.py:207-212) reads it."""
    state = {}
    # a skipped step does not advance torch's Adam either; `applied`: the count of an earlier moment (a Snapshot being written)
    step = float(eng.applied_steps() if applied is None else applied)
    for i, (k, p) in enumerate(model.named_parameters()):
        off, n = eng.layout[k]
        state[i] = {'step': torch.tensor(step), 'exp_avg': eng.adam_m[off:off + n].view(p.shape).cpu().clone(),
                    'exp_avg_sq': eng.adam_v[off:off + n].view(p.shape).cpu().clone()}
    group = {'lr': args.lr, 'betas': (args.beta, 0.999), 'eps': 1e-8, 'weight_decay': 0, 'amsgrad': False, 'maximize': False,
             'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None, 'decoupled_weight_decay': False,
             'params': list(range(len(state)))}
    return {'state': state, 'param_groups': [group]}


def evaluate(model, eng, args, val=None, gap=None):
    """main_train.py:377-438: forward over the validation samples -- CustomDataSet(frame_gap=test_gap): sample k is entry
    k * test_gap, floor(N / test_gap) of them -- PSNR per frame, decoder FPS.  val: (frames, embeds) when the validation
    samples are not a subset of the resident training frames (frame_gap > 1)."""
    frames, embeds = val if val is not None else (eng.frames, eng.embeds)
    gap = args.test_gap if gap is None else gap
    idx = list(range(frames.shape[0])) if val is not None else [k * gap for k in range(frames.shape[0] // gap)]
    psnrs = []
    torch.cuda.synchronize()
    t0 = time.time()
    for k in idx:
        img = eng.decode(embeds[k])
        stats, _ = ops.loss_stats(img, frames[k:k + 1], 'L2', want_grad=False)
        psnrs.append(stats[4])
    torch.cuda.synchronize()
    dt = time.time() - t0
    ms = [utils.msssim_fn([eng.decode(embeds[k])], [frames[k:k + 1]])[0, 0] for k in idx]  # untimed
    return float(torch.stack(psnrs).mean()), len(psnrs) / dt, float(torch.stack(ms).mean())


def train(args):
    """One independent fit per video; the videos of the job are dealt round-robin to the ranks (dist_utils.shard_videos), each
    into its own output directory, so ranks never write the same file.  Returns {video name: best train PSNR} of this rank."""
    rank, local, world = du.env_world()
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    dist = du.init(getattr(args, 'dist_backend', None))
    videos = video_list(args, world)
    mine = du.shard_videos(len(videos), world, rank)
    best, frames_done, steps_done = {}, 0.0, 0.0
    start = time.time()
    base_outf = args.outf
    for v in mine:
        args.outf = video_outf(base_outf, videos, v)
        b, f, st = fit_video(args, videos[v], v, rank)
        best[videos[v]] = b
        frames_done, steps_done = frames_done + f, steps_done + st
    args.outf = base_outf
    secs = time.time() - start
    on_gpu = dist is not None and dist.get_backend() == 'nccl'
    recs = du.gather_records(dist, [sum(best.values()), float(len(best)), frames_done, secs, steps_done],
                             device=f'cuda:{torch.cuda.current_device()}' if on_gpu else 'cpu')
    if rank == 0:
        agg = du.aggregate(recs, max(r[3] for r in recs))
        print(f'Training complete in {secs:.1f}s: {agg}', flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return best


FALLBACK = {'fp16': 'bf16', 'bf16': 'fp32'}


def fit_video(args, name, vid_index, rank, _inject=None):
    """One fit.  _inject(epoch, eng): test hook, called after the epoch's start-of-epoch snapshot."""
    torch.manual_seed(args.manualSeed)                              # main_train.py:162
    PE = utils.PositionalEncoding(args.embed)
    args.embed_length = PE.embed_length
    model = omodel.Generator(embed_length=args.embed_length, stem_dim_num=args.stem_dim_num, fc_hw_dim=args.fc_hw_dim,
                             expansion=args.expansion, num_blocks=args.num_blocks, norm=args.norm, act=args.act, bias=True,
                             reduction=args.reduction, conv_type=args.conv_type, stride_list=args.strides,
                             sin_res=args.single_res, lower_width=args.lower_width, sigmoid=args.sigmoid,
                             deploy=args.deploy, branch_type=args.branch_type)
    total_params = sum(p.numel() for p in model.parameters()) / 1e6
    precision = args.precision
    eng = oeng.TrainEngine(model, loss_type=args.loss_type, beta=args.beta, precision=precision)
    frames, pos = load_frames(args, eng.out_hw, eng.device, name, vid_index, args.frame_gap)
    n = frames.shape[0]
    embeds = PE(pos)
    eng.set_video(frames, embeds)                                   # pos: model.py:37,68
    val = None
    if args.frame_gap != 1:                                         # the validation samples are not a subset of the training ones
        vf, vp = load_frames(args, eng.out_hw, eng.device, name, vid_index, args.test_gap)
        val = (vf, PE(vp).to(eng.device))
    os.makedirs(args.outf, exist_ok=True)
    log = open(os.path.join(args.outf, f'rank{rank}.txt'), 'a')

    def say(msg, console=True):
        if console:
            print(msg, flush=True)
        print(msg, file=log, flush=True)
    say(f'{args}\n Model Params: {total_params}M', console=False)
    g = torch.Generator()
    best = Best()
    best_snap, best_dirty = None, False
    start = time.time()
    steps_per_epoch = min(n, 11) if args.debug else n
    skipped_before = 0
    epoch = 0
    while epoch < args.epochs:
        epoch_start = Snapshot(eng, epoch) if precision in FALLBACK else None
        if _inject is not None:
            _inject(epoch, eng)
        g.manual_seed(args.manualSeed + epoch)
        order = torch.randperm(n, generator=g).tolist()[:steps_per_epoch]
        # Adam's step numbers: the device subtracts the steps THIS engine skipped; steps skipped by an engine a fall-back replaced
        # (eng.skipped_carry) are taken off here, so the bias corrections count applied steps only, whatever the history
        entries = [(f, epoch * steps_per_epoch + i + 1 - eng.skipped_carry, utils.lr_value(epoch % args.epochs, i, n, args))
                   for i, f in enumerate(order)]
        eng.set_schedule(entries)
        eng.run(len(entries))
        st = eng.stats(len(entries))                                 # syncs once per epoch
        sc = eng.scale_state()                                       # (the stats read above has already synchronised)
        warn = skipped_steps_warning(skipped_before, sc['skipped'], len(entries), precision, sc['scale'])
        if warn and precision in FALLBACK:
            # The fit has left this precision's number range: skipped steps change nothing, so it
            # would burn the remaining epochs.  Go back to the start of this epoch and continue in the next wider
            # precision -- same kernels built for bf16 (fp32's exponent range), then the fp32 engine; no restart.
            wider = FALLBACK[precision]
            say(warn.split(' Re-run')[0] + f' Restoring the start of epoch {epoch + 1} and continuing in --precision {wider}.')
            epoch_start.restore(eng)
            step0 = eng.global_step
            carry = step0 - epoch_start.applied                          # steps skipped before the restored epoch began, by any engine
            m0, v0 = eng.adam_m.clone(), eng.adam_v.clone()
            del eng
            eng = oeng.TrainEngine(model, loss_type=args.loss_type, beta=args.beta, precision=wider)
            eng.adam_m.copy_(m0); eng.adam_v.copy_(v0)
            eng.global_step = step0
            eng.skipped_carry = carry                                   # the new engine's own counter starts at 0
            eng.set_video(frames, embeds)
            precision, skipped_before = wider, 0
            best_snap = None if best_snap is None else best_snap       # (arena-sized tensors: still valid for the new engine)
            continue
        skipped_before = sc['skipped']
        if warn:
            say(warn)
        train_psnr = st[:, 4].mean()
        is_train_best = bool(train_psnr > best.train_psnr)            # main_train.py:271-272
        if is_train_best:
            best.train_psnr = train_psnr
            best_snap, best_dirty = Snapshot(eng, epoch), True
        line = (f'[{time.strftime("%Y/%m/%d %H:%M:%S")}] Rank:{rank}, Video:{name}, Epoch[{epoch + 1}/{args.epochs}], lr:{st[-1, 5]:.2e} '
                f'PSNR: {train_psnr:.2f}, best: {float(best.train_psnr):.2f}, loss: {st[:, 0].mean():.5f}, '
                f'{(time.time() - start) / (epoch + 1):.3f} s/epoch')
        say(line, console=epoch % max(1, args.print_freq // 10) == 0 or epoch == args.epochs - 1)
        is_eval = (epoch + 1) % args.eval_freq == 0 or epoch > args.epochs - 10     # main_train.py:303
        is_val_best = False
        if is_eval:
            val_psnr, fps, val_msssim = evaluate(model, eng, args, val)
            is_val_best = val_psnr > float(best.val_psnr)               # main_train.py:310-312
            if is_val_best:
                best.val_psnr = torch.tensor(val_psnr)
            best.val_msssim = torch.maximum(best.val_msssim, torch.tensor(val_msssim))
            # Train MS-SSIM: the reference averages msssim_fn over the epoch's steps (main_train.py:253,258,273); it is a
            # logging metric kept off the timed step here, so it is measured on the eval epochs, after the epoch's updates:
            # on the training frames, which ARE the validation samples unless --frame_gap / --test_gap thin them out.
            tr_msssim = val_msssim if (val is None and args.test_gap == 1) else evaluate(model, eng, args, None, gap=1)[2]
            best.train_msssim = torch.maximum(best.train_msssim, torch.tensor(tr_msssim))
            say(f'Eval best_PSNR at epoch{epoch + 1}:\tcurrent: {val_psnr:.2f}\tbest: {float(best.val_psnr):.2f} \tbest_msssim: '
                f'{float(best.val_msssim):.4f}\t MS-SSIM {val_msssim:.4f} train MS-SSIM {tr_msssim:.4f} decode FPS {fps:.1f}')
        write_now = (args.ckpt_freq and (epoch + 1) % args.ckpt_freq == 0) or (not args.ckpt_freq and is_eval) or epoch == args.epochs - 1
        if is_val_best:                                                  # main_train.py:320-321: the train-mode file only
            save_checkpoint(args, model, eng, epoch, best, 'model_val_best.pth', deploy_too=False)
        if write_now:
            save_checkpoint(args, model, eng, epoch, best)               # model_latest(.pth|_deploy.pth)
            if best_dirty:                                               # model_train_best(.pth|_deploy.pth): the best EPOCH's state
                best_snap.write(args, model, eng, best, 'model_train_best.pth')
                best_dirty = False
        epoch += 1
    torch.cuda.synchronize()
    if args.branch_type == 'ERB':                                        # main_train.py:361-367
        from . import checkpoint
        say(f'Deploy Rep-Model Params: {checkpoint.deploy_param_count(model) / 1e6:.3f}M')
    sc = eng.scale_state()
    if sc['skipped']:
        say(f'loss scale: {sc["skipped"]} steps skipped (non-finite gradients), scale now {sc["scale"]:g}', console=False)
    if precision != args.precision:
        say(f'precision: started in {args.precision}, finished in {precision}')
    say(f'Training complete in: {time.time() - start:.1f}s', console=False)
    log.close()
    return float(best.train_psnr), float(args.epochs * steps_per_epoch), float(eng.global_step)


def main(argv=None):
    train(parse_args(argv))


if __name__ == '__main__':
    main()
