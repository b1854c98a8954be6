# fp32-engine step time for each library given: tools/probes/fp32_variants.sh lib1.so lib2.so ..
for l in "$@"; do ORN_LIB_PATH=$PWD/$l python bench.py --precision fp32 --no-cpu-baseline --quick --steps 40 --warmup 8 --burn-in 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$l'.split('/')[-1], '%.3f ms/step  %.1f f/s' % (d['ms_per_step'], d['value']), ' | '.join('%s %.0f us %.1f TF' % (k['kernel'][:28], k['us_per_step'], k['tflops']) for k in d['roofline']['kernels'][:3]))"; done
