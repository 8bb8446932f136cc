#!/usr/bin/env python3
"""Per-(kernel, grid) table of a rocprofv3 rocpd kernel trace: the same kernel at different problem sizes shows up as
separate rows (dev tool).   python tools/trace_by_shape.py <results.db> [top] [nsteps]"""
import collections
import re
import sqlite3
import sys

db = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 120
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
gcols = [k for k in ("grid_size_x", "grid_size_y", "grid_size_z", "workgroup_size_x") if k in cols]
if not gcols:
    gcols = [k for k in cols if "grid" in k.lower() or "workgroup" in k.lower()][:4]
print("# columns:", cols)
rows = c.execute(f"select name, start, end, {', '.join(gcols)} from kernels order by start").fetchall()
# step marker: the mapping network's pixel-norm runs once per E_align step (first kernel of the G forward)
marks = [i for i, r in enumerate(rows) if "pixelnorm_kernel" in r[0]]
if len(marks) >= nsteps + 1:
    rows = rows[marks[-nsteps - 1]: marks[-1]]
else:
    nsteps = 1
agg = collections.OrderedDict()
for r in rows:
    name = re.sub(r"\(.*", "", r[0].replace("(anonymous namespace)::", "")).replace("void ", "")
    name = name.replace("unsigned short", "bf16")
    key = (name[:80],) + tuple(r[3:])
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
tot = sum(a[1] for a in agg.values())
print(f"# window: last {nsteps} steps, {len(rows)} dispatches = {len(rows) / nsteps:.0f} per step, kernel time {tot / nsteps / 1e3:.3f} ms per step")
print(f"{'n/step':>7} {'us/step':>9} {'avg_us':>8}  {'grid':>22}  name")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{a[0] / nsteps:7.1f} {a[1] / nsteps:9.1f} {a[1] / a[0]:8.1f}  {str(k[1:]):>22}  {k[0]}")
