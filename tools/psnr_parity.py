#!/usr/bin/env python3
"""PSNR parity of the engine's precision modes: train the BASELINE config-2 model (720p, ERB, 9_16_26) with the reference
recipe (lr 5e-4, warm-up 0.2, cosine to zero, Adam(0.5, 0.999), b = 1, shuffled epochs) on the same synthetic video from the
same seeds in each mode and compare the final eval PSNR -- the reference's own metric (main_train.py:377-438: decode every
frame, mean of the per-frame PSNRs).  One JSON line per (seed, precision) + a summary line."""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from orn_amd import ops


def run(precision, epochs, frames, seed, noise, kind, cutoff=20.0):
    import orn_amd.data as data_mod
    data_mod.TEXTURE_CUTOFF = cutoff
    bench.CFG['frames'] = frames
    bench.CFG['epochs'] = epochs
    bench.CFG['warmup'] = int(0.2 * epochs)
    # 'fp32p': the fp32 engine from an initial state that differs by ONE ulp in one weight -- the control that shows how far
    # two equally exact fp32 trainings drift apart on this content (trajectory chaos, not precision)
    eng = bench.make_engine(seed=1234 + seed, precision='fp32' if precision == 'fp32p' else precision, noise=noise, kind=kind, init_seed=1 + seed)
    if precision == 'fp32p':
        off, _ = eng.layout['stem.0.weight']
        eng.params[off:off + 1] = torch.nextafter(eng.params[off:off + 1], torch.full((1,), 10.0, device=eng.params.device))
    n = frames
    t0 = time.time()
    hist = []
    for ep in range(epochs):
        eng.set_schedule(bench.schedule(n, start_step=ep * n))
        eng.run(n)
        hist.append(float(eng.stats(n)[:, 4].mean()))
    torch.cuda.synchronize()
    dt = time.time() - t0
    ps = []
    for k in range(n):
        img = eng.decode(eng.embeds[k])
        s, _ = ops.loss_stats(img, eng.frames[k:k + 1], 'L2', want_grad=False)
        ps.append(float(s[4]))
    sc = eng.scale_state()
    return dict(seed=seed, precision=precision, eval_psnr=sum(ps) / len(ps), train_psnr_last_epoch=hist[-1], seconds=dt,
                fps=epochs * n / dt, skipped=sc['skipped'], scale=sc['scale'], train_psnr_curve=[round(h, 2) for h in hist[::max(1, epochs // 10)]])


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--epochs', type=int, default=30)
    ap.add_argument('--frames', type=int, default=132)
    ap.add_argument('--precisions', default='fp32,fp16')
    ap.add_argument('--noise', type=float, default=0.0)
    ap.add_argument('--kind', default='texture')
    ap.add_argument('--seeds', type=int, default=3)
    ap.add_argument('--cutoff', type=float, default=20.0, help='texture band limit, cycles per image height at 1 sigma')
    a = ap.parse_args()
    precs = a.precisions.split(',')
    res = {}
    for seed in range(a.seeds):
        for p in precs:
            r = run(p, a.epochs, a.frames, seed, a.noise, a.kind, a.cutoff)
            res[(seed, p)] = r
            print(json.dumps(r), flush=True)
    base = precs[0]
    for p in precs[1:]:
        d = [res[(s, p)]['eval_psnr'] - res[(s, base)]['eval_psnr'] for s in range(a.seeds)]
        print(json.dumps({'summary': f'{p} - {base}', 'eval_psnr_delta_dB_per_seed': [round(x, 4) for x in d],
                          'mean_delta_dB': sum(d) / len(d), 'mean_abs_delta_dB': sum(abs(x) for x in d) / len(d),
                          'config': vars(a)}), flush=True)
