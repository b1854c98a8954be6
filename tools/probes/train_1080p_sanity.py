"""BASELINE config 3 geometry (1080p, 9_16_48, strides 5 3 2 2 2) on the fp16 engine: 12 epochs over 24 synthetic frames;
train PSNR must rise and stay finite."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
bench.CFG['frames'] = 24
bench.CFG['epochs'] = 12
bench.CFG['warmup'] = 2
eng = bench.make_engine(seed=1234, precision='fp16', fc_hw_dim='9_16_48', strides=[5, 3, 2, 2, 2], hw=(1080, 1920), frames=24)
t0 = time.time()
for ep in range(12):
    eng.set_schedule(bench.schedule(24, start_step=ep * 24))
    eng.run(24)
    st = eng.stats(24)
    print(f'epoch {ep + 1} train PSNR {float(st[:, 4].mean()):.2f} dB loss {float(st[:, 0].mean()):.4f} finite {bool(torch.isfinite(st).all())}  {time.time() - t0:.1f} s', flush=True)
