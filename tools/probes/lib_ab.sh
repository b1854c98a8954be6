# whole-step A/B of library builds x environment on one box.  usage: lib_ab.sh "<lib.so> [ENV=.. ...] [bench flags]" ...   (two rounds each)
for r in 1 2; do for a in "$@"; do
  envs=""; flags=""; lib=""; for w in $a; do case $w in *.so) lib=$(realpath $w);; *=*) envs="$envs $w";; *) flags="$flags $w";; esac; done
  echo "round $r [$a]"; env ORN_LIB_PATH=$lib $envs python bench.py --no-cpu-baseline --no-fp32 --quick --steps 264 --warmup 66 $flags 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.4f ms/step  %.1f f/s' % (d['ms_per_step'], d['value']))"; done; done
