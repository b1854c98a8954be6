"""Throughput of the other BASELINE shapes on the fp16 engine: config 3 geometry (1080p, 9_16_48, strides 5 3 2 2 2),
NeRV_vanilla 720p, and decode-only FPS at 720p (main_train.py:428's metric)."""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench


def train_fps(eng, n_frames, steps=132, warm=33):
    bench.CFG['frames'] = n_frames
    eng.set_schedule(bench.schedule(warm + steps))
    eng.run(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(steps)
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)


out = {}
eng = bench.make_engine(seed=1234, precision='fp16', fc_hw_dim='9_16_48', strides=[5, 3, 2, 2, 2], hw=(1080, 1920), frames=24)
out['1080p_ERB_9_16_48_train_fps'] = train_fps(eng, 24)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(48):
    eng.decode(eng.embeds[k % 24])
torch.cuda.synchronize()
out['1080p_ERB_decode_fps_train_structure'] = 48 / (time.perf_counter() - t0)
del eng; torch.cuda.empty_cache()
bench.CFG['frames'] = 132
eng = bench.make_engine(seed=1234, precision='fp16', branch_type='NeRV_vanilla')
out['720p_NeRV_vanilla_train_fps'] = train_fps(eng, 132)
del eng; torch.cuda.empty_cache()
eng = bench.make_engine(seed=1234, precision='fp16')
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(264):
    eng.decode(eng.embeds[k % 132])
torch.cuda.synchronize()
out['720p_ERB_decode_fps_train_structure'] = 264 / (time.perf_counter() - t0)
print(json.dumps(out))
