"""Run-to-run determinism of the small ERB engine, repeated: which output differs first (loss ring row, parameter block).
usage: determinism.py [precision=bf16] [reps=8] [steps=12]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import orn_amd
from orn_amd import model, engine
from oracle import cpu_ref
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
ref = None
for rep in range(reps):
    torch.manual_seed(1)
    gen = model.Generator(embed_length=80, stem_dim_num='32_1', fc_hw_dim='2_3_26', expansion=1, num_blocks=1, norm='none', act='swish',
                          bias=True, reduction=2, conv_type='conv', stride_list=[5, 2, 2], sin_res=True, lower_width=96, sigmoid=False,
                          deploy=False, branch_type='ERB')
    eng = engine.TrainEngine(gen, loss_type='Fusion6', beta=0.5, precision=prec)
    hw = eng.out_hw
    frames = cpu_ref.synthetic_video(4, hw[0], hw[1], seed=5)
    embeds = cpu_ref.positional_encoding(torch.tensor([k / 4 for k in range(4)]), 1.25, 40)
    eng.set_video(frames, embeds)
    eng.set_schedule([(k % 4, k + 1, 5e-4) for k in range(steps)])
    eng.run(steps, graph=True)
    torch.cuda.synchronize()
    out = (eng.params.clone(), eng.stats(steps).clone(), {k: v for k, v in eng.layout.items()})
    if ref is None:
        ref = out
    else:
        st_bad = [i for i in range(steps) if not torch.equal(ref[1][i], out[1][i])]
        blocks = [k for k, (off, n) in out[2].items() if not torch.equal(ref[0][off:off + n], out[0][off:off + n])]
        print(f'rep {rep}: first differing stats row {st_bad[:1]}, differing parameter blocks {len(blocks)} {blocks[:4]}', flush=True)
    del eng, gen
    junk = torch.full((64 * 1024 * 1024,), float(rep + 1), device='cuda')
    del junk
