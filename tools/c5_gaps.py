#!/usr/bin/env python3
"""Development probe: idle gaps of the GPU in a rocprofv3 kernel trace (union of all kernels' [start, end]) -- python tools/c5_gaps.py <results.db> [min_gap_us]
Prints busy / idle totals of the last 40 % of the trace and the largest gaps with the kernels either side."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
t0, t1 = rows[0][1], max(r[2] for r in rows)
lo = t0 + 0.6 * (t1 - t0)
rows = [r for r in rows if r[1] >= lo]
short = lambda n: n.split("(")[0].replace("_ZN4isac", "")[:44]
end, last, busy, gaps = rows[0][1], rows[0][0], 0.0, []
for n, a, b in rows:
    if a > end:
        gaps.append(((a - end) / 1e3, short(last), short(n)))
        busy += 0
        end, last = b, n
    if b > end:
        end, last = b, n
span = (max(r[2] for r in rows) - rows[0][1]) / 1e3
idle = sum(g[0] for g in gaps)
print(f"span {span / 1e3:.2f} ms, idle {idle / 1e3:.2f} ms in {len(gaps)} gaps ({100 * idle / span:.1f} %), gaps >= {thr} us: {sum(1 for g in gaps if g[0] >= thr)} totalling {sum(g[0] for g in gaps if g[0] >= thr) / 1e3:.2f} ms")
agg = {}
for g, a, b in gaps:
    if g >= thr:
        k = (a, b); c = agg.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += g
for (a, b), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {n:4d} x  {tot / n:8.1f} us  after {a:46s} before {b}")
