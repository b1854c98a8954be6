"""Median kernel duration per GEMM grid and debug flag from a rocprofv3 db of tools/probes/merge_probe.py."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
flags = [int(x) for x in sys.argv[2].split(',')]
rows = list(db.cursor().execute("select name,start,end,grid_x/workgroup_x,grid_y,grid_z from kernels where name like 'k_gemm_f32%' order by start"))
seq = defaultdict(list)
for r in rows:
    seq[(r[3], r[4], r[5])].append((r[2] - r[1]) / 1e3)
for k, v in seq.items():
    n = len(v) // len(flags)
    out = []
    for i, f in enumerate(flags):
        ch = sorted(v[i * n:(i + 1) * n])
        out.append(f"{f}:{ch[len(ch) // 2]:.1f}")
    print(k, n, ' '.join(out))
