"""Host-side cost of a call on an idle stream: time until run() returns vs until the device is done, per launch mode
(usage: launch_cost.py [stream|graph|eager]).  A host that needs as long to enqueue a step as the device needs to run it is the bound."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
mode = {'stream': None, 'graph': True, 'eager': False}[sys.argv[1] if len(sys.argv) > 1 else 'stream']
eng = bench.make_engine(seed=1234, precision='fp16')
eng.set_schedule(bench.schedule(2000))
eng.run(66, graph=mode); torch.cuda.synchronize()
for n in (1, 4, 20, 20, 20, 100):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.run(n, graph=mode); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'n={n}: host returns after {1e3*(t1-t0):.3f} ms, device done after {1e3*(t2-t0):.3f} ms = {1e3*(t2-t0)/n:.4f} ms/step', flush=True)
