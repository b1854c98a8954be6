# whole-step A/B of the launch modes (and of probe switches) on one box.  usage: mode_ab.sh ["ENV=.. --mode m" ...]; default: the three modes
if [ $# -eq 0 ]; then set -- "--mode graph" "--mode eager" "--mode stream"; fi
for r in 1 2; do for a in "$@"; do
  envs=""; flags=""; for w in $a; do case $w in *=*) envs="$envs $w";; *) flags="$flags $w";; esac; done
  echo "round $r [$a]"; env $envs python bench.py --no-cpu-baseline --no-fp32 --quick --steps 264 --warmup 66 $flags 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.4f ms/step  %.1f f/s' % (d['ms_per_step'], d['value']))"; done; done
