#!/usr/bin/env python3
"""Print the kernel timeline of the last frame in a rocprofv3 rocpd database (kernel-trace run of bench.py):
per-kernel totals, the frame's span, and the time no kernel at all was running (launch gaps)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'k_raygen' in r[0]]
last = rows[idx[-1]:]
t0 = last[0][1]
agg = {}
busy_end, idle, span_end = t0, 0, t0
for n, s, e in last:
    short = n.split('(')[0].replace('rtx::', '')[:40]
    if (s - t0) / 1e6 > 200: break
    gap = max(0, s - busy_end)
    idle += gap
    busy_end = max(busy_end, e)
    span_end = max(span_end, e)
    if len(sys.argv) > 2: print("%8.3f %8.3f gap %6.3f %s" % ((s - t0) / 1e6, (e - s) / 1e6, gap / 1e6, short))
    if short.startswith('__amd') or short.startswith('void at'): continue
    agg.setdefault(short, [0, 0.0]); agg[short][0] += 1; agg[short][1] += (e - s) / 1e6
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("%-28s n=%3d total %8.3f ms" % (k, c, t))
print("frame span %.3f ms, of which no kernel running %.3f ms" % ((span_end - t0) / 1e6, idle / 1e6))
