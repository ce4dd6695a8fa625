"""Per-kernel totals from a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`):
    python tools/kernel_stats.py gpurun_out/xyz/*_results.db [top N]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = con.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                   "from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print("kernel,calls,total_ms,avg_us,min_us,max_us,share")
for r in rows[:top]:
    print('"%s",%d,%.3f,%.2f,%.2f,%.2f,%.4f' % (r[0][:110], r[1], r[2], r[3], r[4], r[5], r[2] / total))
