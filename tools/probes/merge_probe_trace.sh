cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/mp
rocprofv3 --kernel-trace -d /tmp/mp -o k --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/probes/merge_probe.py fwd > /tmp/mp.log 2>&1
tail -5 /tmp/mp.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/mp/**/*kernel_trace.csv', recursive=True)[0]
seq = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_gemm' in r['Kernel_Name']:
        key = (r['Kernel_Name'][:28], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
        seq[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in seq.items():
    v = sorted(v); print(k, len(v), 'median %.1f us' % v[len(v) // 2])
PY
