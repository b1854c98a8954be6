"""Run-to-run bit-identity of 40 training steps at 720p (fresh engines, polluted allocator in between)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
outs = []
for prec in ('fp16', 'fp16', 'bf16', 'bf16'):
    eng = bench.make_engine(seed=1234, precision=prec, noise=0.0)
    eng.set_schedule(bench.schedule(40))
    eng.run(40)
    torch.cuda.synchronize()
    outs.append((prec, eng.params.clone(), eng.stats(40)[:, 0].clone()))
    del eng
    junk = torch.full((256 * 1024 * 1024,), 3.0, device='cuda'); del junk
    torch.cuda.empty_cache()
for a, b in ((0, 1), (2, 3)):
    print(outs[a][0], 'params equal:', torch.equal(outs[a][1], outs[b][1]), 'loss equal:', torch.equal(outs[a][2], outs[b][2]),
          'first diff step:', int((outs[a][2] != outs[b][2]).float().argmax()) if not torch.equal(outs[a][2], outs[b][2]) else -1)
