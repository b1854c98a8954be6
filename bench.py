#!/usr/bin/env python3
"""bench.py -- training frames/sec of the Online-RepNeRV hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 it either runs under
torch.distributed.run (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or, started
without WORLD_SIZE, launches its own N ranks (fresh child processes, one per device, spawned before this process touches
the GPU) and relays rank 0's line.  A "step" is one optimiser step on one frame (main_train.py:229-254: stem,
5 x {online ERB merge, conv3x3+PixelShuffle+SiLU}, head, Fusion6 loss + PSNR, backward, Adam) of BASELINE config 2
(Bunny-shaped 132x1280x720 synthetic video, ERB, fc_hw_dim 9_16_26, strides 5 2 2 2 2) with the video already
resident in HBM.  Independent per-video fits shard one per rank with no data-path collective ("scaling": "weak"); RCCL
carries only the barrier, the max-over-ranks time and the per-rank rates.  Rank 0 prints ONE JSON line.

The headline `value` is the engine's 16-bit mode (--precision fp16: IEEE-half activations and MFMA operands, fp32
accumulate, fp32 master weights / merge / loss / Adam, dynamic loss scale).  The reference trains in fp32, so the same
line carries the fp32 engine's own record under "fp32" (same step, same video, exact-fp32 MFMA), measured in this run.

Launch form (--mode, reported as config.launch_mode): "stream" (default) enqueues the steps through orn_engine_train_steps --
plain launches; in the 16-bit modes the last block's weight gradient -> slab reduction -> merge backward -> Adam -> next merge
forward chain of every step runs on a second stream the engine owns, beside the boundary between that step and the next
 --, "graph" replays the serial step as a hipGraph (rounds 1-3), "eager" calls orn_engine_train_step per step.
All three compute the same bits.  The per-kernel roofline table is measured on serial eager steps (one kernel at a time).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG = dict(embed='1.25_40', stem_dim_num='512_1', fc_hw_dim='9_16_26', expansion=1, reduction=2, lower_width=96,
           strides=[5, 2, 2, 2, 2], frames=132, lr=5e-4, epochs=300, warmup=60, beta=0.5, loss='Fusion6')
CONFIGS = {
    # BASELINE config 2 (the metric's config) and config 3's geometry (optional line, --config 1080p)
    '720p': dict(fc_hw_dim='9_16_26', strides=[5, 2, 2, 2, 2], hw=(720, 1280), flop_step=605.65e9,
                 name='configs[1]: Bunny-shaped 132x3x720x1280 synthetic video, branch_type=ERB, fc_hw_dim 9_16_26, strides 5 2 2 2 2'),
    '1080p': dict(fc_hw_dim='9_16_48', strides=[5, 3, 2, 2, 2], hw=(1080, 1920), flop_step=1366.56e9,
                  name='configs[2] geometry: 132x3x1080x1920 synthetic video, branch_type=ERB, fc_hw_dim 9_16_48, strides 5 3 2 2 2'),
}
PEAK = {'fp32': 157.3, 'bf16': 2500.0, 'fp16': 2500.0}       # TFLOP/s: fp32 MFMA / dense 16-bit MFMA (MI355X_MICROARCH.md)
# kernel symbols as rocprofv3 prints them (template arguments: waves M x N, wave tile M x N, epilogue, K chunk, all-taps)
SYM = {'fwd_last': 'k_conv_fwd_nhwc_bf16<4, 2, 2, 2, 3, 96, false>', 'fwd': 'k_conv_fwd_nhwc_bf16<4, 2, 2, 2, 0, 96, false>',
       'fwd_narrow': 'k_conv_fwd_nhwc_bf16<4, 2, 2, 2, 0, 32, true>',
       # >= 128 pixel tiles: the two-work-groups-per-CU family (orn_conv2_bf16.hip; template argument 0 dgrad, 2 forward of the last block)
       'fwd_last_big': 'k_conv2_nhwc<2>', 'fwd_big': 'k_conv2_nhwc<1>', 'dgrad': 'k_conv2_nhwc<0>',
       'dgrad_split': 'k_conv_nhwc_bf16<8, 1, 1, 3, 2, 96, false> + k_dgrad_finish',
       'dgrad_narrow': 'k_conv_nhwc_bf16<8, 1, 1, 1, 2, 96, true>', 'wgrad': 'k_wgrad_nhwc_bf16_all',
       'wgrad_reduce': 'k_wgrad_bf16_reduce_all'}


def layer_geo(cfg):
    fc_h, fc_w, C = (int(x) for x in cfg['fc_hw_dim'].split('_'))
    H, W = fc_h, fc_w
    out = []
    for i, s in enumerate(cfg['strides']):
        new = int(C * CFG['expansion']) if i == 0 else max(C // CFG['reduction'], CFG['lower_width'])
        out.append(dict(C=C, O=new * s * s, s=s, H=H, W=W, flops=2.0 * C * 9 * new * s * s * H * W))
        C, H, W = new, H * s, W * s
    return out


def make_engine(seed, precision, cfg=None, branch_type='ERB', noise=0.1, fc_hw_dim=None, strides=None, hw=None, frames=None,
                kind='waves', init_seed=1):
    """Engine + resident synthetic video.  cfg: an entry of CONFIGS (default 720p), or the geometry given piecewise."""
    import torch
    cfg = dict(cfg or CONFIGS['720p'])
    if fc_hw_dim:
        cfg.update(fc_hw_dim=fc_hw_dim, strides=strides, hw=hw)
    from orn_amd import engine, model, ops
    from orn_amd.data import synthetic_video
    torch.manual_seed(init_seed)                            # main_train.py:162 (seed 1)
    gen = model.Generator(embed_length=80, stem_dim_num=CFG['stem_dim_num'], fc_hw_dim=cfg['fc_hw_dim'],
                          expansion=CFG['expansion'], num_blocks=1, norm='none', act='swish', bias=True,
                          reduction=CFG['reduction'], conv_type='conv', stride_list=cfg['strides'], sin_res=True,
                          lower_width=CFG['lower_width'], sigmoid=False, deploy=False, branch_type=branch_type)
    eng = engine.TrainEngine(gen, loss_type=CFG['loss'], beta=CFG['beta'], precision=precision)
    n = frames or CFG['frames']
    frames = synthetic_video(n, cfg['hw'][0], cfg['hw'][1], seed=seed, device=eng.device, noise=noise, kind=kind)
    pos = torch.tensor([float(k) / n for k in range(n)], dtype=torch.float32)
    embeds = ops.pe_forward(pos.to(eng.device), 1.25, 40)
    eng.set_video(frames, embeds)
    return eng


def schedule(n_steps, start_step=0):
    """Shuffled-epoch frame order + the reference LR schedule (utils.py:240-259)."""
    import torch
    from orn_amd import utils

    class A:
        lr, epochs, warmup, lr_type, lr_steps = CFG['lr'], CFG['epochs'], CFG['warmup'], 'cosine', []
    n = CFG['frames']
    out = []
    g = torch.Generator()
    for i in range(n_steps):
        step = start_step + i
        epoch, it = divmod(step, n)
        if it == 0 or i == 0:
            g.manual_seed(1 + epoch)
            order = torch.randperm(n, generator=g).tolist()
        out.append((order[it], step + 1, utils.lr_value(epoch % CFG['epochs'], it, n, A)))
    return out


def first_fast_layer(geo, precision):
    """The engine's rule (orn_engine.hip first_fast_layer): trailing layers with C == 96 on the 16-bit kernels, plus one
    narrower layer below them zero-padded to 96 channels."""
    if precision == 'fp32':
        return len(geo)
    ff = len(geo)
    while ff > 0 and geo[ff - 1]['C'] == 96 and geo[ff - 1]['O'] % 96 == 0:
        ff -= 1
    if 0 < ff < len(geo) and geo[ff - 1]['C'] < 96 and geo[ff - 1]['O'] % 96 == 0:
        ff -= 1
    return ff


def conv_kernels(eng, precision, cfg, iters=20):
    """Durations of the conv launches measured LIVE inside real training steps: `iters` eager steps with HIP events around
    every forward conv, every dgrad launch, the batched wgrad launch and its reduction, on the launch stream
    (orn_engine_profile_step) -- the same launches rocprofv3 averages per symbol.  Returns the per-symbol table
    [{kernel, launches_per_step, us_per_step, gflop_per_step, tflops, frac}] (algorithmic FLOPs: real channels only)."""
    geo = layer_geo(cfg)
    nl = len(geo)
    ff = first_fast_layer(geo, precision)
    eng.set_schedule(schedule(iters + 2))
    acc = None
    for k in range(iters + 2):
        ms = eng.profile_step()
        if k >= 2:
            flat = ms['fwd'] + ms['dgrad'] + [ms['wgrad'], ms['wgrad_reduce']]
            acc = flat if acc is None else [a + b for a, b in zip(acc, flat)]
    us = [a / iters * 1e3 for a in acc]
    fwd, dgr, wg, wr = us[:nl], us[nl:2 * nl], us[2 * nl], us[2 * nl + 1]
    rows = {}

    def add(key, t_us, fl):
        r = rows.setdefault(key, dict(kernel=key, launches_per_step=0, us_per_step=0.0, gflop_per_step=0.0))
        r['launches_per_step'] += 1
        r['us_per_step'] += t_us
        r['gflop_per_step'] += fl / 1e9

    if precision == 'fp32':
        for i, L in enumerate(geo):
            add('k_conv3x3_f32<EPI_PS_SILU> (forward, L0..L%d)' % (nl - 1), fwd[i], L['flops'])
            add('fp32 backward call per layer (dgrad + wgrad + dbias kernels)', dgr[i], 2 * L['flops'])
    else:
        ns = 'orn_f16::' if precision == 'fp16' else 'orn_bf16::'
        for i in range(ff, nl):
            L = geo[i]
            narrow = L['C'] <= 32
            tiles = ((L['W'] + 31) // 32) * ((L['H'] + 7) // 8)
            big_last = i == nl - 1 and tiles >= 128 and L['O'] % 96 == 0                       # orn_launch_conv_bf16_fwd
            big_mid = i < nl - 1 and tiles >= 400 and L['O'] % 96 == 0 and not narrow          # (blocks that write the activation copy)
            add(ns + (SYM['fwd_last_big'] if big_last else SYM['fwd_last'] if i == nl - 1 else SYM['fwd_big'] if big_mid else (SYM['fwd_narrow'] if narrow else SYM['fwd'])),
                fwd[i], L['flops'])
            small = tiles < 128 and L['O'] // 96 > 1                                            # orn_launch_conv_bf16_dgrad
            if i == ff:
                add(ns + (SYM['dgrad_narrow'] if (small and narrow) else SYM['dgrad_split']) + ' (fp32 hand-off)', dgr[i], L['flops'])
            else:
                add(ns + (SYM['dgrad_split'] if small else SYM['dgrad']), dgr[i], L['flops'])
        add(ns + SYM['wgrad'] + ' (all 16-bit layers in one launch)', wg, sum(geo[i]['flops'] for i in range(ff, nl)))
        add(ns + SYM['wgrad_reduce'] + ' (split-K slab reduction: HBM-bound, no MFMA work)', wr, 0.0)
    out = []
    for r in rows.values():
        r['tflops'] = r['gflop_per_step'] / r['us_per_step'] * 1e3 if r['us_per_step'] > 0 else 0.0
        r['frac'] = r['tflops'] / PEAK[precision]
        out.append(r)
    out.sort(key=lambda r: -r['us_per_step'])
    return out


def csrc_hash():
    """sha256 over the kernel sources (csrc/*.hip, *.h, sorted by name): identifies the BUILD a measurement belongs to."""
    import glob
    import hashlib
    h = hashlib.sha256()
    pkg = glob.glob(os.path.join(ROOT, 'boosting-neural-video-representation-via-online-structural-reparameteration_amd', 'csrc'))[0]
    for f in sorted(os.listdir(pkg)):
        if f.endswith('.hip') or f.endswith('.h'):
            h.update(f.encode())
            h.update(open(os.path.join(pkg, f), 'rb').read())
    return h.hexdigest()[:16]


def traffic_for(kernel):
    """HBM bytes per launch from the committed PMC pass (profiles/conv_traffic.json, collected as MI355X_MICROARCH.md
    prescribes: FETCH_SIZE and WRITE_SIZE in separate --pmc runs, reads doubled for wide coalesced loads).  The file names
    the kernel symbol it was measured on AND the hash of the kernel sources it was measured with: a symbol that is not the
    one priced here, or a file from another build (round 2: the slab configuration changed under an unchanged symbol),
    is REFUSED (None)."""
    path = os.path.join(ROOT, 'profiles', 'conv_traffic.json')
    if not os.path.exists(path):
        return None
    doc = json.load(open(path))
    if doc.get('csrc_hash') != csrc_hash():
        return None
    want = kernel.split('::', 1)[-1].split(' (')[0].split(' + ')[0]           # template name without namespace / notes
    for rec in doc.get('kernels', []):
        if rec.get('symbol', '').split('::', 1)[-1].startswith(want):
            return rec.get('traffic_bytes_per_launch')
    return None


def usable_cores():
    """Cores this process may really use: the affinity mask, cut to the cgroup's CPU quota when there is one (the 1-GPU box
    shows 256 hardware threads to os.cpu_count() and grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(steps=4):
    """The CPU oracle ("port": same ATen CPU ops as the reference's CPU path) timed on this host, bounded sample: per thread
    setting 1 warm-up + `steps` ERB training steps at 720p with Fusion6 (all host cores and 8 threads).
    The headline entry is the all-cores one."""
    import torch
    from oracle import cpu_ref
    sd = cpu_ref.init_state_dict(80, CFG['stem_dim_num'], CFG['fc_hw_dim'], CFG['strides'], CFG['expansion'],
                                 CFG['reduction'], CFG['lower_width'], 'ERB', seed=1)
    am = {k: torch.zeros_like(v) for k, v in sd.items()}
    av = {k: torch.zeros_like(v) for k, v in sd.items()}
    frames = cpu_ref.synthetic_video(2, 720, 1280, seed=1234)
    embeds = cpu_ref.positional_encoding(torch.tensor([0.0, 1.0 / 132]), 1.25, 40)
    usable = usable_cores()
    runs = []
    step = 0
    for threads in sorted({usable, min(8, usable)}, reverse=True):
        torch.set_num_threads(threads)
        step += 1
        cpu_ref.train_step(sd, am, av, step, 5e-5, embeds[0:1], frames[0:1], CFG['fc_hw_dim'], CFG['strides'], 'ERB', 'Fusion6', 0.5)
        t0 = time.time()
        for i in range(steps):
            step += 1
            cpu_ref.train_step(sd, am, av, step, 5e-5, embeds[i % 2:i % 2 + 1], frames[i % 2:i % 2 + 1], CFG['fc_hw_dim'],
                               CFG['strides'], 'ERB', 'Fusion6', 0.5)
        runs.append({'threads': threads, 'value': steps / (time.time() - t0), 'unit': 'frames/s'})
    return dict(value=runs[0]['value'], unit='frames/s', cores=runs[0]['threads'], kind='port',
                sample=f'{steps} ERB 720p training steps (Fusion6, Adam) after 1 warm-up per thread setting, oracle/cpu_ref.py on torch-CPU',
                runs=runs, cpu_model=cpu_model(), os_cpu_count=os.cpu_count(), usable_cores=usable)


class StubEngine:
    """--cpu-stub (tests of the launcher / reduction path on machines without a GPU): a step is a short sleep."""

    def set_schedule(self, s):
        self.n = len(s)

    def run(self, n, graph=None):
        time.sleep(0.002 * n)

    def stats(self, n):
        import torch
        return torch.zeros(n, 8)


def burn_in(eng, n, graph=None):
    """Setup, not warm-up: n replays of the step on the freshly built engine, after which parameters, Adam moments, step count and
    loss scale are put back, so the W warm-up and K timed steps that follow are exactly the steps they would have been without it.
    The first ~100 steps of a process run 2 % slower than the rest (device clocks and first touches; same box: --steps 20 --warmup 5
    1.100 ms vs --steps 264 --warmup 66 1.079 ms per step), a fit is 39,600 steps long, and the driver's timed region is 20 steps
    after 5: without this the line reports the start-up transient instead of the rate the hot path sustains.  Reported as
    `burn_in_steps` in the JSON line; `--burn-in 0` switches it off."""
    import torch
    if n <= 0:
        return
    keep = (eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng.global_step)
    sc = eng.scale_state()
    eng.set_schedule(schedule(n))
    eng.run(n, graph=graph)
    torch.cuda.synchronize()
    eng.params.copy_(keep[0]); eng.adam_m.copy_(keep[1]); eng.adam_v.copy_(keep[2])
    eng.global_step = keep[3]
    eng.set_grad_scale(sc['scale'], sc['ceiling'])
    torch.cuda.synchronize()


def timed_leg(eng, steps, warmup, graph, dist, device_sync):
    """W untimed + K timed steps, barrier + synchronize on both sides, MAX over ranks.  Returns (dt_max, dt_own, stats)."""
    import torch
    eng.set_schedule(schedule(warmup + steps))
    eng.run(warmup, graph=graph)
    p0 = eng.params.clone() if hasattr(eng, 'params') else None     # (witness below; outside the timed region)
    device_sync()
    if dist:
        dist.barrier()
    device_sync()
    t0 = time.perf_counter()
    eng.run(steps, graph=graph)
    device_sync()
    own = time.perf_counter() - t0
    if dist:
        dist.barrier()
    device_sync()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda' if torch.cuda.is_available() and dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if p0 is not None:     # liveness witness of the timed steps (after the clock has stopped): how far they moved the parameters
        eng.last_param_delta_l2 = float((eng.params.double() - p0.double()).norm())
        del p0
    return dt, own, eng.stats(warmup + steps)


def worker(args):
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    stub = args.cpu_stub
    if os.environ.get('ORN_BENCH_FAIL_RANK') == str(rank):          # tests: a rank that dies before the rendezvous
        raise SystemExit(3)
    if not stub:
        torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # "nccl" = RCCL over xGMI; device_id binds the communicator to this rank's GPU at creation (no reliance on the current device)
        kw = {'device_id': torch.device('cuda', local)} if (args.backend == 'nccl' and not stub) else {}
        dist.init_process_group(args.backend, rank=rank, world_size=world, **kw)
    device_sync = (lambda: None) if stub else torch.cuda.synchronize
    cfg = CONFIGS[args.config]
    graph = {'stream': None, 'graph': True, 'eager': False}['eager' if args.no_graph else args.mode]

    eng = StubEngine() if stub else make_engine(seed=1234 + rank, precision=args.precision, cfg=cfg)   # one independent video per rank
    if not stub:
        burn_in(eng, args.burn_in, graph)
    dt, own, stats = timed_leg(eng, args.steps, args.warmup, graph, dist, device_sync)
    psnr_last = float(stats[args.warmup:, 4].mean())
    ok = bool(torch.isfinite(stats[:, 0]).all())
    per_rank = [args.steps / own]
    if dist:
        lst = [None] * world
        dist.all_gather_object(lst, args.steps / own)
        per_rank = [float(x) for x in lst]

    if rank == 0:
        out = {
            'metric': 'training frames/sec, Bunny 720p ERB', 'value': world * args.steps / dt, 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'burn_in_steps': 0 if stub else args.burn_in,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'fp32': 'f32', 'bf16': 'bf16', 'fp16': 'f16'}[args.precision], 'data': 'synthetic',
            'config': {'workload': cfg['name'] + ', stem 512_1, lower_width 96, Fusion6, Adam(0.5,0.999), b=1; one independent video per GPU',
                       'precision': args.precision, 'hip_graph': bool(graph),
                       'launch_mode': 'eager' if args.no_graph else args.mode,
                       'timed_region': 'optimiser steps incl. loss, backward, Adam, per-step PSNR; MS-SSIM logging excluded, checkpoint I/O excluded',
                       'train_psnr_mean_timed_steps': psnr_last, 'finite': ok,
                       # witnesses that the timed steps trained (they differ between precisions and builds): loss of the last timed step,
                       # L2 norm of (parameters after - before the timed region), Adam's step count as the device wrote it into the ring
                       'last_timed_step_loss': float(stats[-1, 0]), 'last_timed_step_adam_count': int(stats[-1, 7]),
                       'param_delta_l2_timed_steps': getattr(eng, 'last_param_delta_l2', None),
                       'whole_step_tflops': world * args.steps / dt * cfg['flop_step'] / 1e12},
            'rccl_ranks': (dist.get_world_size() if dist else 1), 'per_rank_frames_per_s': per_rank,
        }
        if args.config != '720p':
            out['metric'] = f'training frames/sec, {args.config} ERB (not the BASELINE metric config)'
        if not stub:
            sc = eng.scale_state()
            out['config']['loss_scale'] = {'scale': sc['scale'], 'steps_skipped': sc['skipped'], 'halvings': sc['backoffs']}
            kern = conv_kernels(eng, args.precision, cfg)
            dom = next(k for k in kern if k['gflop_per_step'] > 0)   # dominant by time (among the kernels that do MFMA work)
            step_us = dt / args.steps * 1e6
            out['roofline'] = {
                'bound': 'mfma', 'achieved': dom['tflops'], 'peak': PEAK[args.precision], 'unit': 'TFLOP/s', 'frac': dom['frac'],
                'traffic': traffic_for(dom['kernel']) if args.config == '720p' and args.precision == 'fp16' else None,   # the PMC file is the 720p fp16 step
                'kernel': dom['kernel'] + f" ({dom['launches_per_step']} launches/step; dominant by time)",
                'flops_per_launch': dom['gflop_per_step'] * 1e9 / dom['launches_per_step'],
                'avg_launch_ms': dom['us_per_step'] / dom['launches_per_step'] / 1e3,
                'kernels': kern,
                'whole_step_frac': (cfg['flop_step'] / (step_us * 1e-6) / 1e12) / PEAK[args.precision],
                'conv_us_per_step': sum(k['us_per_step'] for k in kern), 'step_us': step_us,
            }
            if world == 1 and args.config == '720p' and not args.quick:
                # SURVEY 8(d)'s definition of the metric: wall clock over >= 3 full epochs after 1 warm-up epoch (outside `value`'s
                # timed region; `value` keeps the driver's --steps / --warmup)
                n = CFG['frames']
                dts, _, sts = timed_leg(eng, 3 * n, n, graph, None, device_sync)
                out['sustained'] = {'value': 3 * n / dts, 'unit': 'frames/s', 'steps': 3 * n, 'warmup': n, 'ms_per_step': dts / (3 * n) * 1e3,
                                    'definition': '3 full epochs of the 132-frame video after 1 warm-up epoch',
                                    'finite': bool(torch.isfinite(sts[:, 0]).all())}
            del eng
            torch.cuda.empty_cache()
            if world == 1 and args.config == '720p' and not args.quick:
                # BASELINE config 3's geometry (1080p, fc_hw_dim 9_16_48, strides 5 3 2 2 2) on 12 resident frames, same precision
                c3 = CONFIGS['1080p']
                keep = CFG['frames']
                CFG['frames'] = 12
                try:
                    e3 = make_engine(seed=1234, precision=args.precision, cfg=c3, frames=12)
                    dt3, _, st3 = timed_leg(e3, 66, 24, graph, None, device_sync)
                    k3 = conv_kernels(e3, args.precision, c3, iters=8)
                    d3 = next(k for k in k3 if k['gflop_per_step'] > 0)
                    out['cfg3_1080p'] = {'workload': c3['name'] + ' (12 resident frames)', 'value': 66 / dt3, 'unit': 'frames/s', 'steps': 66, 'warmup': 24,
                                         'ms_per_step': dt3 / 66 * 1e3, 'finite': bool(torch.isfinite(st3[:, 0]).all()),
                                         'roofline': {'frac': d3['frac'], 'achieved': d3['tflops'], 'peak': PEAK[args.precision], 'unit': 'TFLOP/s',
                                                      'kernel': d3['kernel'], 'avg_launch_ms': d3['us_per_step'] / d3['launches_per_step'] / 1e3},
                                         'whole_step_frac': (66 / dt3 * c3['flop_step'] / 1e12) / PEAK[args.precision]}
                    del e3
                finally:
                    CFG['frames'] = keep
                torch.cuda.empty_cache()
            if not args.no_fp32 and args.precision != 'fp32' and world == 1:
                # the reference's own arithmetic: the same step on the exact-fp32 engine, measured in this run
                e32 = make_engine(seed=1234 + rank, precision='fp32', cfg=cfg)
                s32 = min(args.steps, args.fp32_steps)
                w32 = min(args.warmup, 12)
                dt32, _, st32 = timed_leg(e32, s32, w32, graph, None, device_sync)
                k32 = conv_kernels(e32, 'fp32', cfg, iters=6)
                d32 = k32[0]
                out['fp32'] = {
                    'value': s32 / dt32, 'unit': 'frames/s', 'steps': s32, 'warmup': w32, 'ms_per_step': dt32 / s32 * 1e3, 'dtype': 'f32',
                    'finite': bool(torch.isfinite(st32[:, 0]).all()), 'train_psnr_mean_timed_steps': float(st32[w32:, 4].mean()),
                    'last_timed_step_loss': float(st32[-1, 0]), 'last_timed_step_adam_count': int(st32[-1, 7]),
                    'param_delta_l2_timed_steps': getattr(e32, 'last_param_delta_l2', None),
                    'whole_step_tflops': s32 / dt32 * cfg['flop_step'] / 1e12,
                    'roofline': {'bound': 'mfma', 'achieved': d32['tflops'], 'peak': PEAK['fp32'], 'unit': 'TFLOP/s', 'frac': d32['frac'],
                                 'kernel': d32['kernel'], 'kernels': k32,
                                 'whole_step_frac': (s32 / dt32 * cfg['flop_step'] / 1e12) / PEAK['fp32']},
                }
                del e32
                torch.cuda.empty_cache()
            if not args.no_cpu_baseline and world == 1 and args.config == '720p':
                out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def launch(args):
    """`--gpus N` without an outer launcher: N fresh child processes, one rank per device.  This process has not touched the
    GPU (no torch.cuda call before this point) and never execs; a failing rank makes the whole run fail."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile
    logdir = tempfile.mkdtemp(prefix='orn_bench_ranks_')
    procs, logs = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   ORN_BENCH_CHILD='1')
        # ranks >= 1 print nothing on stdout in a good run; what they do print (and their stderr) goes to a per-rank log that is
        # shown when the run fails, instead of being thrown away
        log = None if r == 0 else open(os.path.join(logdir, f'rank{r}.log'), 'w')
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=(subprocess.PIPE if r == 0 else log), stderr=(None if r == 0 else subprocess.STDOUT), text=True))
    # rank 0's line is read by a thread; the ranks are polled so that one dying rank ends the run (its peers would wait in
    # the rendezvous or a barrier forever): the children this process started are then terminated by PID
    import threading
    buf = []
    th = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    th.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
        time.sleep(0.05)
    if failed is None:
        failed = next((r for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write(f'bench.py: rank {failed} failed (exit codes {[p.returncode for p in procs]})\n')
        for r, log in enumerate(logs):
            if log is not None:
                log.close()
                tail = open(log.name).read()[-2000:]
                if tail.strip():
                    sys.stderr.write(f'--- rank {r} output (tail of {log.name}) ---\n{tail}\n')
        raise SystemExit(1)
    th.join(timeout=10)
    sys.stdout.write(buf[0] if buf else '')
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=264)
    ap.add_argument('--warmup', type=int, default=66)
    ap.add_argument('--burn-in', type=int, default=132, dest='burn_in',
                    help='setup replays of the step before the warm-up, state restored afterwards (see burn_in); 0 = off')
    ap.add_argument('--precision', default=os.environ.get('ORN_PRECISION', 'fp16'), choices=['fp32', 'bf16', 'fp16'])
    ap.add_argument('--config', default='720p', choices=sorted(CONFIGS))
    ap.add_argument('--fp32-steps', type=int, default=66, help='timed steps of the fp32 record (at most --steps)')
    ap.add_argument('--no-fp32', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--mode', default='stream', choices=['stream', 'graph', 'eager'],
                    help='stream (default): orn_engine_train_steps, plain launches, the last block\'s weight-gradient chain pipelined on the '
                         'engine\'s second stream; graph: hipGraph replay of the serial step; eager: one orn_engine_train_step call per step')
    ap.add_argument('--no-graph', action='store_true', help='same as --mode eager')
    ap.add_argument('--quick', action='store_true', help='headline leg and its roofline only (probes): no sustained / 1080p legs')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'])
    ap.add_argument('--cpu-stub', action='store_true', help='no GPU: exercise launcher + reduction only (tests)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch(args)
    worker(args)


if __name__ == '__main__':
    main()
