#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace:  python tools/prof_summary.py <results.db> [--csv out.csv]
Prints per-kernel count / avg / total / share, like `rocprofv3 --stats` does for csv output."""
import sqlite3
import sys


def summarise(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc")
    rows = list(cur.execute(q))
    tot = sum(r[3] for r in rows) or 1
    return [(r[0], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[3] / tot) for r in rows]


def summarise_pmc(path):
    """Per-kernel average of every collected counter (rocprofv3 --pmc ...)."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pm = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    ip = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    q = (f"select s.kernel_name, i.name, count(*), avg(p.value), avg(d.end-d.start)/1e3 from {pm} p join {kd} d on p.event_id=d.event_id "
         f"join {ks} s on d.kernel_id=s.id join {ip} i on p.pmc_id=i.id group by s.kernel_name, i.name order by 4 desc")
    return list(cur.execute(q))


def overlap(path):
    """Concurrency analysis of a kernel trace: union busy time vs sum of durations, and for every kernel which other
    kernels were resident while it ran (time-weighted).  Uses the middle 60 % of the trace (steady state)."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo, hi = t0 + 0.2 * (t1 - t0), t0 + 0.8 * (t1 - t0)
    rows = [(n.split("(")[0][:60], max(a, lo), min(b, hi)) for n, a, b in rows if b > lo and a < hi]
    ev = sorted([(a, 1, n) for n, a, b in rows] + [(b, -1, n) for n, a, b in rows])
    active, last, busy, hist, co = {}, lo, 0.0, {}, {}
    for t, d, n in ev:
        dt = t - last
        if dt > 0:
            k = sum(active.values())
            hist[k] = hist.get(k, 0.0) + dt
            if k:
                busy += dt
                for a in active:
                    if active[a]:
                        c = co.setdefault(a, {})
                        for b in active:
                            if active[b] and b != a:
                                c[b] = c.get(b, 0.0) + dt
                        c["__self__"] = c.get("__self__", 0.0) + dt
        active[n] = active.get(n, 0) + d
        last = t
    total = sum(b - a for _, a, b in rows)
    print(f"window {1e-6 * (hi - lo):.2f} ms: busy (union) {1e-6 * busy:.2f} ms, sum of kernel durations {1e-6 * total:.2f} ms, "
          f"mean concurrency while busy {total / max(busy, 1):.2f}")
    print("resident-kernel histogram: " + "  ".join(f"{k}:{100 * v / (hi - lo):.0f}%" for k, v in sorted(hist.items())))
    for a, c in sorted(co.items(), key=lambda kv: -kv[1]["__self__"])[:8]:
        self_t = c.pop("__self__")
        tops = sorted(c.items(), key=lambda kv: -kv[1])[:4]
        print(f"  {a[:44]:44s} resident {1e-6 * self_t:8.2f} ms; with: " + ", ".join(f"{b[:28]} {100 * v / self_t:.0f}%" for b, v in tops))


def gaps(path):
    """Idle-gap attribution of a kernel trace (middle 60 %): every interval with NO kernel resident, which kernel ended last before it
    and which started first after it."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    import re

    def short(n):
        m = re.search(r"\d+([a-z][a-z0-9_]*_kernel)", n.replace("isac", ""))
        return (m.group(1) if m else n)[:28]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo, hi = t0 + 0.2 * (t1 - t0), t0 + 0.8 * (t1 - t0)
    rows = [(short(n), a, b) for n, a, b in rows if b > lo and a < hi]
    cur_end, last_name, out = None, None, []
    for n, a, b in rows:
        if cur_end is not None and a > cur_end:
            out.append((a - cur_end, last_name, n))
        if cur_end is None or b > cur_end:
            cur_end, last_name = b, n
    tot = sum(g[0] for g in out)
    print(f"window {1e-6 * (hi - lo):.2f} ms: {len(out)} idle gaps, {1e-6 * tot:.2f} ms idle ({100 * tot / (hi - lo):.1f} %), mean {1e-3 * tot / max(len(out), 1):.1f} us")
    edges = [2e3, 5e3, 10e3, 20e3, 50e3, 100e3, 1e12]
    acc = [[0, 0.0] for _ in edges]
    for g, _, _ in out:
        for i, e in enumerate(edges):
            if g < e:
                acc[i][0] += 1; acc[i][1] += g
                break
    print("gap length histogram (count, share of idle): " + "  ".join(f"<{int(e / 1e3) if e < 1e11 else 'inf'}us: {c} ({100 * t / max(tot, 1):.0f}%)" for e, (c, t) in zip(edges, acc)))
    pairs = {}
    for g, a, b in out:
        k = (a, b)
        pairs[k] = (pairs.get(k, (0, 0.0))[0] + 1, pairs.get(k, (0, 0.0))[1] + g)
    for (a, b), (c, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  after {a:28s} before {b:28s} n={c:4d} idle {1e-6 * t:7.2f} ms  mean {1e-3 * t / c:6.1f} us")


def timeline(path, span_ms=3.5):
    """Start / end of every kernel launch inside a few milliseconds in the middle of the trace (relative microseconds, one line per launch,
    queue id when the trace has one): what actually runs beside what."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    q = "d.queue_id" if "queue_id" in cols else "0"
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, {q} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    import re

    def short(n):
        m = re.search(r"\d+([a-z][a-z0-9_]*_kernel)", n.replace("isac", ""))
        return (m.group(1) if m else n)[:26]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    mid = t0 + 0.7 * (t1 - t0)
    lo, hi = mid, mid + span_ms * 1e6
    for n, a, b, qid in rows:
        if b > lo and a < hi:
            print(f"{1e-3 * (a - lo):9.1f} {1e-3 * (b - lo):9.1f}  {1e-3 * (b - a):7.1f} us  q{qid}  {short(n)}")


if __name__ == "__main__":
    if "--timeline" in sys.argv:
        timeline(sys.argv[1])
        sys.exit(0)
    if "--gaps" in sys.argv:
        gaps(sys.argv[1])
        sys.exit(0)
    if "--overlap" in sys.argv:
        overlap(sys.argv[1])
        sys.exit(0)
    if "--pmc" in sys.argv:
        rows = summarise_pmc(sys.argv[1])
        lines = ["kernel,counter,dispatches,avg_value,avg_duration_us"]
        for r in rows:
            lines.append(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.1f},{r[4]:.2f}")
        if "--csv" in sys.argv:
            open(sys.argv[sys.argv.index("--csv") + 1], "w").write("\n".join(lines) + "\n")
        for r in rows[:16]:
            print(f"{r[0][:70]:70s} {r[1]:12s} n={r[2]:4d} avg={r[3]:14.1f} dur={r[4]:9.1f}us")
        sys.exit(0)
    rows = summarise(sys.argv[1])
    lines = ["kernel,calls,avg_us,total_us,min_us,max_us,pct"]
    for r in rows:
        lines.append(f"\"{r[0]}\",{r[1]},{r[2]:.2f},{r[3]:.2f},{r[4]:.2f},{r[5]:.2f},{r[6]:.2f}")
    if "--csv" in sys.argv:
        open(sys.argv[sys.argv.index("--csv") + 1], "w").write("\n".join(lines) + "\n")
    for r in rows[:24]:
        print(f"{r[0][:84]:84s} n={r[1]:5d} avg={r[2]:9.1f}us tot={r[3]:10.1f}us {r[6]:5.1f}%")
