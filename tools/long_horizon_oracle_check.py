"""Manual check (not collected by pytest; needs a GPU and ~10 minutes): does the REFERENCE recipe itself leave the
~40 dB plateau on the clean synthetic 720p video, or only the engine?

Trains BASELINE config 2 with the fp32 engine under the 300-epoch schedule until an epoch collapses (min per-step
PSNR < 25 dB), rewinds to the snapshot taken at the start of the previous epoch (parameters + Adam state), and
replays the same schedule entries on the CPU oracle (reference math, torch autograd + Adam).  Prints both per-step
PSNR traces side by side."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tools/ -> repo root
sys.path.insert(0, ROOT)
import torch
import bench
from oracle import cpu_ref

max_oracle_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 220
bench.CFG['epochs'] = 300
bench.CFG['warmup'] = 60
eng = bench.make_engine(seed=1234, precision='fp32', noise=0.0)
n = 132
snaps = []
armed = False                     # the collapse only counts once the fit has reached its plateau
t0 = time.time()
for ep in range(120):
    sched = bench.schedule(n, start_step=ep * n)
    snaps.append((ep, eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone(), sched))
    snaps = snaps[-2:]
    eng.set_schedule(sched)
    eng.run(n)
    st = eng.stats(n).clone()
    if ep % 10 == 9:
        print(f'# engine epoch {ep + 1} PSNR mean {float(st[:, 4].mean()):.2f} min {float(st[:, 4].min()):.2f}  {time.time() - t0:.0f} s', flush=True)
    armed = armed or float(st[:, 4].mean()) > 35.0
    if armed and (float(st[:, 4].min()) < 25.0 or not torch.isfinite(st[:, 4]).all()):
        break
else:
    print('no collapse within 120 epochs')
    sys.exit(0)
bad = int(((st[:, 4] < 25.0) | ~torch.isfinite(st[:, 4])).nonzero()[0])
print(f'# engine: epoch {ep + 1} first step with PSNR < 25 dB: it {bad} (global step {ep * n + bad + 1}); lr {float(st[bad, 5]):.3e}', flush=True)
# rewind: start of this epoch if the collapse is late in it, else the previous epoch
use_prev = bad < 100 and len(snaps) == 2
ep0, P0, M0, V0, sched0 = snaps[0] if use_prev else snaps[1]
entries = list(sched0)
eng_trace = None
if use_prev:
    # the engine's own trace over the previous epoch + this one: rerun from the snapshot (deterministic)
    entries = list(sched0) + list(snaps[1][4])
eng.params.copy_(P0); eng.adam_m.copy_(M0); eng.adam_v.copy_(V0)
k_total = min(len(entries), (n if use_prev else 0) + bad + 12, max_oracle_steps)
first = max(0, ((n if use_prev else 0) + bad + 12) - k_total)      # oracle budget: replay only the last k_total steps
# advance the engine to `first`, snapshot there, then trace both
if first > 0:
    eng.set_schedule(entries[:first]); eng.run(first); torch.cuda.synchronize()
shapes = {k: tuple(p.shape) for k, p in eng.model.named_parameters()}
cut = lambda arena, k: arena[eng.layout[k][0]:eng.layout[k][0] + eng.layout[k][1]].clone().cpu().view(shapes[k])
sd = {k: cut(eng.params, k) for k in shapes}
am = {k: cut(eng.adam_m, k) for k in shapes}
av = {k: cut(eng.adam_v, k) for k in shapes}
todo = entries[first:first + k_total]
eng.set_schedule(todo); eng.run(len(todo))
tr = eng.stats(len(todo)).clone()
print(f'# replaying {len(todo)} steps from global step {todo[0][1]} on the CPU oracle ({torch.get_num_threads()} threads)', flush=True)
embeds = eng.embeds.cpu(); frames = eng.frames
for i, (f, step, lr) in enumerate(todo):
    loss, psnr, _ = cpu_ref.train_step(sd, am, av, step, lr, embeds[f:f + 1], frames[f:f + 1].cpu(), '9_16_26', [5, 2, 2, 2, 2], 'ERB', 'Fusion6', 0.5)
    print(f'step {step} frame {f} lr {lr:.3e}  engine PSNR {float(tr[i, 4]):7.3f} loss {float(tr[i, 0]):.5f} | oracle PSNR {float(psnr):7.3f} loss {float(loss):.5f}', flush=True)
