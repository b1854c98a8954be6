"""Print the kernel timeline of one training step from a rocprofv3 results .db (kernel trace): start / end relative to the step's
first kernel, so launches that overlap (side branch) show as such.  usage: step_timeline.py <results.db> [steps back from the end, default 12]
(the last steps of a bench.py process are the eager profile steps; 12 back is inside the graph-replayed timed region)"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 12
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
rows = list(c.execute(f"select name,start,end,grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z),workgroup_x{',' + qcol if qcol else ''} from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'k_advance' in r[0]]          # one per step (eager / stream modes; one per graph launch in graph mode)
back = min(back, len(idx) - 2)
a, b = idx[-1 - back - 1] - 1, idx[-1 - back] - 1
t0 = rows[a + 1][1]
busy_end = t0
tot = 0
for r in rows[a + 1:b + 1]:
    nm = re.sub(r'\(.*', '', r[0])
    nm = re.sub('^void ', '', nm)[:56]
    d = (r[2] - r[1]) / 1e3
    ov = '  OVERLAPS' if r[1] < busy_end else ''
    busy_end = max(busy_end, r[2])
    tot += d
    q = f' q={r[5]}' if qcol else ''
    print(f"{nm:56s} wg={r[3]:6d}x{r[4]:4d} {(r[1] - t0) / 1e3:8.1f} -> {(r[2] - t0) / 1e3:8.1f}  {d:7.1f} us{q}{ov}")
print('sum of kernels', round(tot, 1), 'span (advance to advance)', (rows[b + 1][1] - rows[a + 1][1]) / 1e3)
