#!/usr/bin/env python3
"""bench.py -- training frames/sec of the Online-RepNeRV hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under
torch.distributed.run (one rank per GPU, RCCL).  A "step" is one optimiser step on one frame
(main_train.py:229-254: stem, 5x{online ERB merge, conv3x3+PixelShuffle+SiLU}, head, Fusion6 loss +
PSNR, backward, Adam) of BASELINE config 2 (Bunny-shaped 132x1280x720 synthetic video, ERB,
fc_hw_dim 9_16_26, strides 5 2 2 2 2) with the video already resident in HBM.  Independent
per-video fits shard one per rank with no data-path collective ("scaling": "weak"); RCCL is used only
for the barrier / max-over-ranks reduction.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

CFG = dict(embed='1.25_40', stem_dim_num='512_1', fc_hw_dim='9_16_26', expansion=1, reduction=2, lower_width=96,
           strides=[5, 2, 2, 2, 2], frames=132, lr=5e-4, epochs=300, warmup=60, beta=0.5, loss='Fusion6')
FLOP_STEP = 605.65e9          # algorithmic FLOPs per trained 720p frame, 3x fwd


def layer_geo():
    from orn_amd import model  # noqa: F401
    C, H, W = 26, 9, 16
    out = []
    for i, s in enumerate(CFG['strides']):
        new = int(C * CFG['expansion']) if i == 0 else max(C // CFG['reduction'], CFG['lower_width'])
        out.append(dict(C=C, O=new * s * s, s=s, H=H, W=W))
        C, H, W = new, H * s, W * s
    return out


def make_engine(seed, precision, branch_type='ERB', noise=0.1, fc_hw_dim=None, strides=None, hw=(720, 1280), frames=None):
    from orn_amd import engine, model, ops
    from orn_amd.data import synthetic_video
    torch.manual_seed(1)                                    # main_train.py:162
    gen = model.Generator(embed_length=80, stem_dim_num=CFG['stem_dim_num'], fc_hw_dim=fc_hw_dim or CFG['fc_hw_dim'],
                          expansion=CFG['expansion'], num_blocks=1, norm='none', act='swish', bias=True,
                          reduction=CFG['reduction'], conv_type='conv', stride_list=strides or CFG['strides'], sin_res=True,
                          lower_width=CFG['lower_width'], sigmoid=False, deploy=False, branch_type=branch_type)
    eng = engine.TrainEngine(gen, loss_type=CFG['loss'], beta=CFG['beta'], precision=precision)
    n = frames or CFG['frames']
    frames = synthetic_video(n, hw[0], hw[1], seed=seed, device=eng.device, noise=noise)
    pos = torch.tensor([float(k) / n for k in range(n)], dtype=torch.float32)
    embeds = ops.pe_forward(pos.to(eng.device), 1.25, 40)
    eng.set_video(frames, embeds)
    return eng


def schedule(n_steps, start_step=0):
    """Shuffled-epoch frame order + the reference LR schedule (utils.py:240-259)."""
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


def conv_roofline(eng, precision, iters=20):
    """Dominant kernel = the implicit-GEMM 3x3 conv forward.  fp32: k_conv3x3_f32<EPI_PS_SILU>, one launch per layer
    (L0..L4); 16-bit: k_conv_nhwc_bf16<4,2,2,2,EPI_B_FWD_LAST>, the last block's forward (one launch per step,
    152.9 GF at 720p); the earlier fast layers run the <..,EPI_B_FWD> instantiation and are listed in per_layer.  Durations are measured LIVE inside real training steps: the engine
    runs `iters` eager steps with HIP events bracketing every layer's forward conv launch on the launch stream
    (orn_engine_profile_step), so clocks, caches and operands are those of the step -- the same launches rocprofv3
    averages for the symbol.  Returns (algorithmic flops per launch, avg launch duration [s], per-layer list)."""
    geo = layer_geo()
    ff = 0
    if precision in ('bf16', 'fp16'):
        # the engine's rule (orn_engine.hip first_fast_layer): trailing layers with C == 96 and O % 128 == 0, plus one
        # narrower layer below them run zero-padded to 96 channels
        ff = len(geo)
        while ff > 0 and geo[ff - 1]['C'] == 96 and geo[ff - 1]['O'] % 128 == 0:
            ff -= 1
        if 0 < ff < len(geo) and geo[ff - 1]['C'] < 96 and geo[ff - 1]['O'] % 128 == 0:
            ff -= 1
    eng.set_schedule(schedule(iters + 2))
    acc = [0.0] * len(geo)
    for k in range(iters + 2):
        ms = eng.profile_step()
        if k >= 2:
            acc = [a + m for a, m in zip(acc, ms)]
    per_layer, tot_t, tot_f, n = [], 0.0, 0.0, 0
    for li, L in enumerate(geo):
        if li < ff:
            continue
        dt = acc[li] / iters / 1e3
        fl = 2.0 * L['C'] * 9 * L['O'] * L['H'] * L['W']          # algorithmic: the real input channels only
        per_layer.append(dict(layer=li, ms=dt * 1e3, tflops=fl / dt / 1e12))
        tot_t += dt
        tot_f += fl
        n += 1
    if precision in ('bf16', 'fp16'):
        # the last block's forward is its own kernel symbol (<..,EPI_B_FWD_LAST>: no activation copy), one launch per
        # step and 76 % of the forward FLOPs: that launch is the roofline kernel
        last = per_layer[-1]
        L = geo[-1]
        return 2.0 * L['C'] * 9 * L['O'] * L['H'] * L['W'], last['ms'] / 1e3, per_layer
    return tot_f / n, tot_t / n, per_layer


def cpu_baseline(steps=8):
    """The CPU oracle ("port": same ATen CPU ops as the reference's CPU path) timed on this host,
    bounded sample: 1 warm-up + `steps` ERB training steps at 720p with Fusion6."""
    from oracle import cpu_ref
    # the 1-GPU box's CPU share is 16 cores; oneDNN with all 256 hardware threads is pathologically slow
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    sd = cpu_ref.init_state_dict(80, CFG['stem_dim_num'], CFG['fc_hw_dim'], CFG['strides'], CFG['expansion'],
                                 CFG['reduction'], CFG['lower_width'], 'ERB', seed=1)
    am = {k: torch.zeros_like(v) for k, v in sd.items()}
    av = {k: torch.zeros_like(v) for k, v in sd.items()}
    frames = cpu_ref.synthetic_video(2, 720, 1280, seed=1234)
    embeds = cpu_ref.positional_encoding(torch.tensor([0.0, 1.0 / 132]), 1.25, 40)
    cpu_ref.train_step(sd, am, av, 1, 5e-5, embeds[0:1], frames[0:1], CFG['fc_hw_dim'], CFG['strides'], 'ERB', 'Fusion6', 0.5)
    t0 = time.time()
    for i in range(steps):
        cpu_ref.train_step(sd, am, av, 2 + i, 5e-5, embeds[i % 2:i % 2 + 1], frames[i % 2:i % 2 + 1], CFG['fc_hw_dim'],
                           CFG['strides'], 'ERB', 'Fusion6', 0.5)
    dt = time.time() - t0
    return dict(value=steps / dt, unit='frames/s', cores=torch.get_num_threads(), kind='port',
                sample=f'{steps} ERB 720p training steps (Fusion6, Adam) after 1 warm-up, oracle/cpu_ref.py on torch-CPU, '
                       f'{torch.get_num_threads()} threads')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=264)
    ap.add_argument('--warmup', type=int, default=66)
    ap.add_argument('--precision', default=os.environ.get('ORN_PRECISION', 'fp16'), choices=['fp32', 'bf16', 'fp16'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})')
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world)     # RCCL over xGMI

    eng = make_engine(seed=1234 + rank, precision=args.precision)        # one independent video per rank
    sched = schedule(args.warmup + args.steps)
    eng.set_schedule(sched)
    graph = not args.no_graph
    eng.run(args.warmup, graph=graph)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(args.steps, graph=graph)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stats = eng.stats(args.warmup + args.steps)
    psnr_last = float(stats[args.warmup:, 4].mean())
    ok = bool(torch.isfinite(stats[:, 0]).all())

    if rank == 0:
        fl, avg_dt, per_layer = conv_roofline(eng, args.precision)
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'conv_fwd_traffic.json')
        if args.precision != 'fp32' and os.path.exists(tpath):       # PMC pass collected separately (see the file's "method")
            traffic = json.load(open(tpath))['layers'][-1]['traffic_bytes']     # the last block's launch (the roofline kernel)
        peak = 157.3 if args.precision == 'fp32' else 2500.0      # fp32 MFMA / dense 16-bit MFMA (bf16 and f16 share the rate)
        achieved = fl / avg_dt / 1e12
        out = {
            'metric': 'training frames/sec, Bunny 720p ERB', 'value': world * args.steps / dt, 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'fp32': 'f32', 'bf16': 'bf16', 'fp16': 'f16'}[args.precision], 'data': 'synthetic',
            'config': {'workload': 'configs[1]: Bunny-shaped 132x3x720x1280 synthetic video, branch_type=ERB, fc_hw_dim 9_16_26, '
                                   'strides 5 2 2 2 2, stem 512_1, lower_width 96, Fusion6, Adam(0.5,0.999), b=1; '
                                   'one independent video per GPU', 'precision': args.precision, 'hip_graph': graph,
                       'train_psnr_mean_timed_steps': psnr_last, 'finite': ok,
                       'whole_step_tflops': world * args.steps / dt * FLOP_STEP / 1e12},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                         'traffic': traffic,
                         'kernel': ('k_conv3x3_f32<EPI_PS_SILU> (5 launches/step, L0..L4)' if args.precision == 'fp32'
                                    else f'orn_{"bf16" if args.precision == "bf16" else "f16"}::k_conv_nhwc_bf16<4,2,2,2,EPI_B_FWD_LAST> '
                                         f'(1 launch/step: forward conv of the last block, L{per_layer[-1]["layer"]}; per_layer lists the '
                                         f'<..,EPI_B_FWD> launches of L{per_layer[0]["layer"]}..L{per_layer[-2]["layer"]} too)'),
                         'flops_per_launch': fl, 'avg_launch_ms': avg_dt * 1e3, 'per_layer': per_layer},
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
