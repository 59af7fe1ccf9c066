"""Summarise kernel durations from a rocprofv3 rocpd sqlite database (kernel-trace)."""
import sqlite3, sys
for path in sys.argv[1:]:
    db = sqlite3.connect(path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = cur.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc").fetchall()
    print("==", path)
    print("%-90s %6s %10s %10s %10s" % ("kernel", "calls", "avg_us", "min_us", "max_us"))
    for r in rows[:12]:
        print("%-90s %6d %10.2f %10.2f %10.2f" % (r[0][:90], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3))
