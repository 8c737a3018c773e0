#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 results .db (rocpd sqlite): python tools/kstats_db.py <db> [n]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc").fetchall()
tot = sum(r[3] for r in rows)
for name, calls, avg, sm in rows[:n]:
    print(f"{name[:100]:100s} calls={calls:6d} avg_us={avg/1e3:9.1f} tot%={100*sm/tot:5.1f}")
print(f"total kernel time {tot/1e6:.2f} ms")
