# whole-step A/B of environment switches on one box: env_ab.sh "VAR1=1" "VAR2=1" ...   ("" = defaults); two rounds each
for r in 1 2; do for e in "" "$@"; do echo "round $r env [$e]"; env $e python bench.py --no-cpu-baseline --no-fp32 --quick --steps 264 --warmup 66 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.4f ms/step  %.1f f/s | ' % (d['ms_per_step'], d['value']) + ' | '.join('%s %.1f' % (k['kernel'].split('::')[-1][:30], k['us_per_step']) for k in d['roofline']['kernels'][:6]))"; done; done
