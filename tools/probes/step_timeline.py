"""Print the kernel timeline of one training step from a rocprofv3 results .db (kernel trace)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
rows = list(c.execute("select name,start,end,grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z),workgroup_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'k_adam' in r[0]]
a, b = idx[-2], idx[-1]
prev = rows[a][2]
tot = 0
for r in rows[a + 1:b + 1]:
    nm = re.sub(r'\(.*', '', r[0])
    nm = re.sub('^void ', '', nm)[:56]
    d = (r[2] - r[1]) / 1e3
    g = (r[1] - prev) / 1e3
    prev = r[2]
    tot += d
    print(f"{nm:56s} wg={r[3]:6d}x{r[4]:4d} {d:7.1f} us gap {g:5.1f}")
print('sum', tot, 'span', (rows[b][2] - rows[a][2]) / 1e3)
