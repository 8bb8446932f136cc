#!/usr/bin/env python3
"""GPU busy time vs wall time of a kernel trace (rocpd sqlite): union of kernel intervals, idle gaps, and how much of the
busy time sits in short kernels - dev tool.   python tools/gpu_gaps.py <results.db> [window_start_frac window_end_frac]"""
import sqlite3, sys
db = sys.argv[1]
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = sqlite3.connect(db).execute("select start, end, name from kernels order by start").fetchall()
# step markers: every E_align step ends with its second lreq_adam launch; take the window of the last `nsteps` whole steps
adam = [i for i, r in enumerate(rows) if "lreq_adam" in r[2]]
nsteps = int(f1) if len(sys.argv) > 3 else 3
if len(adam) >= 2 * nsteps + 2:
    rows = rows[adam[-2 * nsteps - 1] + 1: adam[-1] + 1]
    print(f"window: last {nsteps} steps")
else:
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo, hi = t0 + (t1 - t0) * f0, t0 + (t1 - t0) * f1
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
busy, cur_s, cur_e, gaps, big = 0, rows[0][0], rows[0][1], [], []
prev_name = rows[0][2]
for s, e, nm in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e)
        if s - cur_e > 20000:
            big.append(((s - cur_e) / 1e3, prev_name.split("(")[0][-60:], nm.split("(")[0][-60:]))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev_name = nm
busy += cur_e - cur_s
wall = max(r[1] for r in rows) - rows[0][0]
short = [r for r in rows if r[1] - r[0] < 10000]
print(f"kernels {len(rows)}, wall {wall/1e6:.2f} ms, busy (union) {busy/1e6:.2f} ms = {100*busy/wall:.1f} %, sum of durations {sum(r[1]-r[0] for r in rows)/1e6:.2f} ms")
print(f"idle gaps: {len(gaps)}, total {sum(gaps)/1e6:.2f} ms; gaps > 20 us: {sum(1 for g in gaps if g > 20000)} totalling {sum(g for g in gaps if g > 20000)/1e6:.2f} ms")
print(f"kernels shorter than 10 us: {len(short)} ({100*len(short)/len(rows):.0f} %), their durations sum to {sum(r[1]-r[0] for r in short)/1e6:.2f} ms")
for g_, a_, b_ in big:
    print(f"  gap {g_:7.1f} us  after {a_}  before {b_}")
import collections, re
cnt = collections.Counter()
dur = collections.Counter()
for s_, e_, n_ in rows:
    k = re.sub(r"\(.*", "", n_.replace("(anonymous namespace)::", "")).replace("void ", "")[:70]
    cnt[k] += 1; dur[k] += e_ - s_
print("launches in the window by kernel (count, total us):")
for k, n in cnt.most_common(28):
    print(f"{n:5d} {dur[k]/1e3:9.1f}  {k}")
