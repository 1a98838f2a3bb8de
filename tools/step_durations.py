#!/usr/bin/env python3
"""Per-step durations from a rocprofv3 kernel trace (rocpd sqlite) of bench.py: python tools/step_durations.py DB [first] [last]
For each launch of the emit kernel: start-to-next-start time, the kernels' own durations in between, and the idle gaps."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
last = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rows = db.execute("select name, start, end from kernels order by start").fetchall()
emits = [i for i, r in enumerate(rows) if "transe_emit" in r[0]]
short = lambda n: n.split("(")[0].split("::")[-1].split("<")[0][:14]
print("step  period_us  busy_us  idle_us  | kernels (us)")
for k in range(first, min(last, len(emits) - 1)):
    seg = rows[emits[k]:emits[k + 1]]
    period = (rows[emits[k + 1]][1] - seg[0][1]) / 1e3
    busy = sum(e - s for _, s, e in seg) / 1e3
    print("%4d  %9.1f  %7.1f  %7.1f  | %s" % (k, period, busy, period - busy, "  ".join("%s %.0f" % (short(n), (e - s) / 1e3) for n, s, e in seg)))
