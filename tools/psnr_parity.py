#!/usr/bin/env python3
"""Train the BASELINE config-2 model for a short schedule in fp32 and bf16 from the same seed on the same
synthetic video and compare train / eval PSNR (the reference's own metric, main_train.py:253-257,377-438)."""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from orn_amd import ops, utils


def run(precision, epochs, frames, seed, noise):
    bench.CFG['frames'] = frames
    bench.CFG['epochs'] = epochs
    bench.CFG['warmup'] = int(0.2 * epochs)
    eng = bench.make_engine(seed=seed, precision=precision, noise=noise)
    n = frames
    t0 = time.time()
    hist = []
    for ep in range(epochs):
        sched = bench.schedule(n, start_step=ep * n)
        eng.set_schedule(sched)
        eng.run(n)
        st = eng.stats(n)
        hist.append(float(st[:, 4].mean()))
        if ep % 20 == 19:
            print(f'# {precision} epoch {ep + 1}/{epochs} train PSNR {hist[-1]:.3f} dB  {time.time() - t0:.0f} s', file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    ps = []
    for k in range(n):
        img = eng.decode(eng.embeds[k])
        s, _ = ops.loss_stats(img, eng.frames[k:k + 1], 'L2', want_grad=False)
        ps.append(float(s[4]))
    return dict(precision=precision, train_psnr_last_epoch=hist[-1], eval_psnr=sum(ps) / len(ps), seconds=dt,
                fps=epochs * n / dt, train_psnr_curve=hist[::max(1, epochs // 10)])


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--epochs', type=int, default=30)
    ap.add_argument('--frames', type=int, default=132)
    ap.add_argument('--precisions', default='fp32,bf16')
    ap.add_argument('--noise', type=float, default=0.1)
    a = ap.parse_args()
    out = [run(p, a.epochs, a.frames, 1234, a.noise) for p in a.precisions.split(',')]
    for o in out:
        print(json.dumps(o), flush=True)
    if len(out) == 2:
        print(json.dumps({'eval_psnr_delta_dB': out[1]['eval_psnr'] - out[0]['eval_psnr'],
                          'train_psnr_delta_dB': out[1]['train_psnr_last_epoch'] - out[0]['train_psnr_last_epoch']}))
