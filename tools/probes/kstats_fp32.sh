# per-kernel time of one fp32-engine training step under rocprofv3.  usage: kstats_fp32.sh <lib.so> <tag> -> gpurun_out/kstats32_<tag>.txt
lib=$1; tag=$2
export ORN_LIB_PATH=$(realpath $lib)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/k32_$tag
rocprofv3 --kernel-trace -d /tmp/k32_$tag -o k --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --precision fp32 --no-cpu-baseline --quick --steps 12 --warmup 4 --burn-in 0 > /tmp/k32_$tag.log 2>&1
python3 - $tag <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/kstats32_$2.txt
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f'/tmp/k32_{tag}/**/*kernel_trace.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    acc[n][0] += 1; acc[n][1] += d
steps = max(v[0] for k, v in acc.items() if 'k_adam' in k)
tot = 0.0; out = []
for k, (c, t) in acc.items():
    per = t / steps
    if per < 2: continue
    out.append((per, c / steps, k)); tot += per
for per, c, k in sorted(out, reverse=True): print(f'{per:8.1f} us/step  x{c:5.1f}  {k[:140]}')
print(f'{tot:8.1f} us/step total ({steps} steps)')
PY
