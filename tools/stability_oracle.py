"""Does the reference recipe (lr 5e-4, warmup 0.2, cosine, Adam(0.5, 0.999), Fusion6) stay stable on the synthetic
video?  Runs the CPU oracle (reference math) on a small geometry with the full 300-epoch schedule shape."""
import sys, os, time, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from oracle import cpu_ref

torch.set_num_threads(8)
fc, strides = sys.argv[1] if len(sys.argv) > 1 else '2_3_26', [5, 2, 2]
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
lr0 = float(sys.argv[4]) if len(sys.argv) > 4 else 5e-4
sd = cpu_ref.init_state_dict(80, '512_1', fc, strides, 1, 2, 96, 'ERB', seed=1)
h = int(fc.split('_')[0]) * 20
w = int(fc.split('_')[1]) * 20
video = cpu_ref.synthetic_video(n, h, w, seed=1234)
emb = cpu_ref.positional_encoding(torch.tensor([k / n for k in range(n)]), 1.25, 40)
m = {k: torch.zeros_like(v) for k, v in sd.items()}
v = {k: torch.zeros_like(x) for k, x in sd.items()}
g = torch.Generator()
step = 0
t0 = time.time()
for ep in range(epochs):
    g.manual_seed(1 + ep)
    order = torch.randperm(n, generator=g).tolist()
    ps = []
    for it, f in enumerate(order):
        step += 1
        lr = cpu_ref.adjust_lr_value(ep, it, n, lr0, epochs, 0.2 * epochs, 'cosine', [])
        loss, psnr, _ = cpu_ref.train_step(sd, m, v, step, lr, emb[f:f + 1], video[f:f + 1], fc, strides)
        ps.append(float(psnr))
    if ep % 10 == 9:
        print(f'epoch {ep + 1} lr {lr:.2e} train PSNR {sum(ps) / len(ps):.3f} dB  {time.time() - t0:.0f} s', flush=True)
