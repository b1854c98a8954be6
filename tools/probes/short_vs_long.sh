# the driver's --steps 20 --warmup 5 with and without the setup burn-in, against the long timed region; same box, alternating
for r in 1 2 3; do
  for a in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --burn-in 0" "--steps 264 --warmup 66 --burn-in 0"; do
    python bench.py --no-cpu-baseline --no-fp32 --quick $a | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$a', '%.4f ms/step  %.1f f/s' % (d['ms_per_step'], d['value']), 'psnr %.3f' % d.get('psnr_last', d.get('config', {}).get('psnr', 0)) if 'psnr_last' in d else '')"
  done
done
