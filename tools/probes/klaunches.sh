# per-LAUNCH durations (median over steps) of the kernels that run more than once per step.  usage: klaunches.sh <pattern> ...
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kl
rocprofv3 --kernel-trace -d /tmp/kl -o k --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32 --steps 40 --warmup 12 $ORN_BENCH_ARGS > /tmp/kl.log 2>&1
python3 - "$@" <<'PY'
import csv, glob, sys, collections
pats = sys.argv[1:]
f = glob.glob('/tmp/kl/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = collections.defaultdict(list)
cnt = collections.Counter()
# position of the launch within its step: count occurrences between k_adam launches
pos = collections.Counter()
for r in rows:
    n = r['Kernel_Name']
    if 'k_adam' in n:
        pos.clear(); continue
    for p in pats:
        if p in n:
            seq[(p, pos[p])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
            pos[p] += 1
for k in sorted(seq):
    v = sorted(seq[k])
    print('%-28s launch %d of the step: median %7.1f us  (n=%d)' % (k[0], k[1], v[len(v) // 2], len(v)))
PY
