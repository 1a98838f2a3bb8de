#!/usr/bin/env python3
"""Timeline of a few steady-state steps from a rocprofv3 kernel trace (rocpd sqlite): python tools/step_timeline.py DB [emit-substring]
Prints, per kernel of two consecutive steps, start / end relative to the first emit kernel's start (us), and the stream (queue)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
key = sys.argv[2] if len(sys.argv) > 2 else "transe_emit_vec_kernel"
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
rows = db.execute("select name, start, end, queue_id from kernels order by start").fetchall() if "kernels" in tabs else []
emits = [i for i, r in enumerate(rows) if key in r[0]]
i0 = emits[len(emits) // 2]
i1 = emits[len(emits) // 2 + 2]
t0 = rows[i0][1]
for name, s, e, q in rows[i0:i1 + 1]:
    print("%9.1f %9.1f  %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name.split("(")[0][-60:]))
