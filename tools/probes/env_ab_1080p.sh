for r in 1 2; do for e in "" "ORN_FWD2_APAD=1"; do echo "round $r env [$e]"; env $e python bench.py --config 1080p --no-cpu-baseline --no-fp32 --steps 132 --warmup 33 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.4f ms/step  %.1f f/s | ' % (d['ms_per_step'], d['value']) + ' | '.join('%s %.1f' % (k['kernel'].split('::')[-1][:30], k['us_per_step']) for k in d['roofline']['kernels'][:5]))"; done; done
