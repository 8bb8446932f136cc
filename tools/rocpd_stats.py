#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table."""
import re
import sqlite3
import sys

db = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end from kernels").fetchall()
agg = {}
for name, s, e in rows:
    short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))
    short = re.sub(r"^void ", "", short)
    a = agg.setdefault(short, [0, 0.0, 1e30, 0.0])
    d = (e - s) / 1e3
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print(f"# kernels: {len(rows)} dispatches, total {tot/1e3:.3f} ms")
print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>9} {'max_us':>10} {'pct':>6}  name")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{a[0]:7d} {a[1]:12.1f} {a[1]/a[0]:10.2f} {a[2]:9.2f} {a[3]:10.2f} {100*a[1]/tot:6.2f}  {k[:150]}")
