"""Long-horizon stability of the engine on a chosen geometry: train PSNR per 10 epochs under the reference schedule
(lr, warmup 0.2, cosine).  Pair with tests/stability_oracle.py (CPU oracle) (same data, same order) on the CPU."""
import sys, os, time, argparse
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from orn_amd import model, engine, ops, utils
from orn_amd.data import synthetic_video

ap = argparse.ArgumentParser()
ap.add_argument('--fc', default='9_16_26'); ap.add_argument('--strides', type=int, nargs='+', default=[5, 2, 2])
ap.add_argument('--epochs', type=int, default=300); ap.add_argument('--frames', type=int, default=16)
ap.add_argument('--lr', type=float, default=5e-4); ap.add_argument('--precision', default='fp32')
ap.add_argument('--noise', type=float, default=0.0); ap.add_argument('--every', type=int, default=10)
a = ap.parse_args()
torch.manual_seed(1)
gen = model.Generator(embed_length=80, stem_dim_num='512_1', fc_hw_dim=a.fc, expansion=1, num_blocks=1, norm='none', act='swish',
                      bias=True, reduction=2, conv_type='conv', stride_list=a.strides, sin_res=True, lower_width=96, sigmoid=False,
                      deploy=False, branch_type='ERB')
eng = engine.TrainEngine(gen, loss_type='Fusion6', beta=0.5, precision=a.precision)
n = a.frames
video = synthetic_video(n, eng.out_hw[0], eng.out_hw[1], seed=1234, device='cpu', noise=a.noise)
emb = ops.pe_forward(torch.tensor([k / n for k in range(n)], dtype=torch.float32).cuda(), 1.25, 40)
eng.set_video(video.cuda(), emb)


class A:
    lr, epochs, warmup, lr_type, lr_steps = a.lr, a.epochs, int(0.2 * a.epochs), 'cosine', []


g = torch.Generator()
t0 = time.time()
for ep in range(a.epochs):
    g.manual_seed(1 + ep)
    order = torch.randperm(n, generator=g).tolist()
    sched = [(f, ep * n + it + 1, utils.lr_value(ep, it, n, A)) for it, f in enumerate(order)]
    eng.set_schedule(sched)
    eng.run(n)
    st = eng.stats(n)
    if ep % a.every == a.every - 1:
        print(f'epoch {ep + 1} lr {sched[-1][2]:.2e} train PSNR {float(st[:, 4].mean()):.3f} dB  loss {float(st[:, 0].mean()):.5f}  {time.time() - t0:.0f} s', flush=True)
