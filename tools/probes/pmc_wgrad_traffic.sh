# HBM traffic of the 16-bit wgrad kernel + its reduce at the four fast-layer shapes (FETCH_SIZE / WRITE_SIZE in separate passes)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_w_$c -o w -- python3 $GRAFT_REPO_ROOT/tools/probes/wgrad_ablate.py 0 > /tmp/pmc_w_$c.log 2>&1
python3 - $c <<'PY'
import sqlite3, sys
c = sys.argv[1]
db = sqlite3.connect(f'/tmp/pmc_w_{c}/w_results.db')
cur = db.cursor()
rows = list(cur.execute("select dispatch_id, name, counter_name, sum(counter_value), max(end - start) from pmc_events where name like '%wgrad%' group by dispatch_id, counter_name order by dispatch_id"))
import collections
seen = collections.OrderedDict()
for did, name, cn, v, dur in rows:
    seen.setdefault((name[:40], round(v * 1024 / 1e6, 1)), []).append(dur)
for (k, mb), v in seen.items():
    print(c, k, mb, 'MB per launch;', len(v), 'launches, mean', round(sum(v) / len(v) / 1e3, 1), 'us')
PY
done
