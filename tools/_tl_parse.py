#!/usr/bin/env python3
"""Development aid: turn the ISAC_TIMELINE=1 lines of a pipelined run (stderr) into a schedule table -- per CPI the start / end of the
beam-sum (B), the fused echo + range kernel (E), the covariance launch group (C) and the end of the CPI (T), microseconds."""
import sys
rows = []
for ln in open(sys.argv[1]):
    if ln.startswith("TL "):
        p = ln.split()
        rows.append((p[1], [float(p[i]) for i in (3, 4, 6, 7, 9, 10, 12)]))
rows = rows[len(rows) // 2:len(rows) // 2 + 14]
t0 = rows[0][1][0]
print("ctx        B0      B1      E0      E1      C0      C1      T  | B    E    C   | E0-prevE1  period(E0)")
prev_e1, prev_e0 = None, None
for c, t in rows:
    t = [x - t0 for x in t]
    print(f"{c[-6:]} " + " ".join(f"{x:7.0f}" for x in t) + f" | {t[1]-t[0]:4.0f} {t[3]-t[2]:4.0f} {t[5]-t[4]:4.0f} | " + (f"{t[2]-prev_e1:6.0f} {t[2]-prev_e0:8.0f}" if prev_e1 is not None else ""))
    prev_e1, prev_e0 = t[3], t[2]
