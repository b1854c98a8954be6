"""Locate the step at which the 720p config-2 fit on the clean synthetic video leaves its ~40 dB plateau."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench

prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
epochs_total, stop = 300, int(sys.argv[2]) if len(sys.argv) > 2 else 62
bench.CFG['epochs'] = epochs_total
bench.CFG['warmup'] = 60
eng = bench.make_engine(seed=1234, precision=prec, noise=0.0)
n = 132
for ep in range(stop):
    eng.set_schedule(bench.schedule(n, start_step=ep * n))
    eng.run(n)
    st = eng.stats(n)
    if ep >= 38:
        ps = st[:, 4]
        bad = (ps < 30).nonzero()
        first = int(bad[0]) if len(bad) else -1
        print(f'epoch {ep + 1} lr {float(st[-1, 5]):.3e} PSNR mean {float(ps.mean()):.2f} min {float(ps.min()):.2f} first<30dB at it {first}', flush=True)
        if first >= 0 and float(ps[0]) > 30:
            lo = max(0, first - 6)
            print('   per-step PSNR around it:', [round(float(x), 2) for x in ps[lo:first + 8]], flush=True)
            print('   per-step loss:', [round(float(x), 5) for x in st[lo:first + 8, 0]], flush=True)
