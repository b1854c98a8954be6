"""A/B of whole-step time between library builds on ONE box (box-to-box spread is 2-4 %): runs bench.py's timed region in a child
process per library, alternating, and prints ms/step of each round.  usage: ab_step.py libA.so libB.so [libC.so ..] [rounds] [steps]"""
import os, subprocess, sys, json
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
libs = [a for a in sys.argv[1:] if a.endswith('.so')]
rest = [a for a in sys.argv[1:] if not a.endswith('.so')]
rounds = int(rest[0]) if len(rest) > 0 else 3
steps = rest[1] if len(rest) > 1 else '264'
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, ORN_LIB_PATH=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--no-cpu-baseline', '--no-fp32', '--quick', '--steps', steps, '--warmup', '66'], env=env,
                             capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[l].append(d['ms_per_step'])
            print(f'round {r} {os.path.basename(l)}: {d["ms_per_step"]:.4f} ms/step  ({d["value"]:.1f} f/s)  conv us: '
                  + ' | '.join(f'{x["kernel"].split("::")[-1][:34]} {x["us_per_step"]:.1f}' for x in d['roofline']['kernels']), flush=True)
        except Exception as e:
            print('FAILED', l, out.stdout[-500:], out.stderr[-1500:], flush=True)
for l in libs:
    if res[l]:
        print(f'{os.path.basename(l)}: best {min(res[l]):.4f}  mean {sum(res[l])/len(res[l]):.4f} ms/step')
