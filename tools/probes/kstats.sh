# per-kernel time of one training step under rocprofv3 for a given library build.  usage: kstats.sh <lib.so> <tag> [bench flags]  -> gpurun_out/kstats_<tag>.txt
# default flags: --mode graph (the SERIAL step: one kernel at a time, so the per-symbol times add up to the step; in --mode stream the side
# branch's launches overlap the caller's stream and the sum exceeds the step: use timeline.sh for that picture)
lib=$1; tag=$2; shift 2; flags="$@"; [ -z "$flags" ] && flags="--mode graph"
export ORN_LIB_PATH=$(realpath $lib)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats -d /tmp/ks_$tag -o k --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32 --quick --steps 40 --warmup 12 $flags > /tmp/ks_$tag.log 2>&1
python3 - $tag <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/kstats_$tag.txt
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f'/tmp/ks_{tag}/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# the timed region: the last 40 steps are graph replays; use all dispatches, normalise per step by counting k_adam launches
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    acc[n][0] += 1; acc[n][1] += d
steps = max(v[0] for k, v in acc.items() if 'k_adam' in k)
tot = 0.0
out = []
for k, (c, t) in acc.items():
    per = t / steps
    if per < 0.5: continue
    out.append((per, c / steps, k))
    tot += per
for per, c, k in sorted(out, reverse=True):
    print(f'{per:8.1f} us/step  x{c:4.1f}  {k[:150]}')
print(f'{tot:8.1f} us/step total ({steps} steps incl. profile/eager steps)')
PY
tail -2 /tmp/ks_$tag.log | cut -c1-300 >> $GRAFT_REPO_ROOT/gpurun_out/kstats_$tag.txt
cp $(find /tmp/ks_$tag -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/kstats_${tag}_rocprof_stats.csv 2>/dev/null
