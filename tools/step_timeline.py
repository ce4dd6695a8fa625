"""Kernel timeline of the last scan steps in a rocprofv3 rocpd database: per kernel name the time inside the
window, and the idle time of the device between consecutive kernels.
    python tools/step_timeline.py db [window_ms 100] [list kernels >= min_us] [skip the last ms]"""
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
rows = con.execute("select name, start, end from kernels order by start").fetchall()
skip = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
t_end = max(r[2] for r in rows) - skip * 1e6
rows = [r for r in rows if t_end - win * 1e6 <= r[1] <= t_end]
busy = defaultdict(float)
calls = defaultdict(int)
gap_after = defaultdict(float)
idle = 0.0
last_end = rows[0][1]
prev = None
for name, s, e in rows:
    if s > last_end:
        idle += s - last_end
        if prev is not None:
            gap_after[prev] += s - last_end
    busy[name] += e - s
    calls[name] += 1
    last_end = max(last_end, e)
    prev = name
span = last_end - rows[0][1]
print(f"window {span / 1e6:.2f} ms, device idle {idle / 1e6:.2f} ms ({idle / span:.1%})")
print("kernel,calls,busy_ms,share_of_window,idle_after_ms")
for name in sorted(busy, key=busy.get, reverse=True)[:30]:
    print('"%s",%d,%.3f,%.4f,%.3f' % (name[:100], calls[name], busy[name] / 1e6, busy[name] / span, gap_after[name] / 1e6))
if len(sys.argv) > 3:                       # the kernels of the window in launch order
    t0 = rows[0][1]
    for name, s, e in rows:
        if e - s >= float(sys.argv[3]) * 1e3:
            print("%9.3f ms  %8.1f us  %s" % ((s - t0) / 1e6, (e - s) / 1e3, name[:120]))
