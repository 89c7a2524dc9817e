#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run stored as SQLite (rocpd): python scripts/prof_db_stats.py out_results.db [n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = db.execute("select name, count(*), avg(end-start)/1000.0, sum(end-start)/1e6, max(end-start)/1000.0 from kernels "
                  "group by name order by 4 desc").fetchall()
print(f"total kernel time {sum(r[3] for r in rows):.2f} ms")
for r in rows[:top]:
    print(f"{r[0][:56]:56s} n={r[1]:6d} avg={r[2]:8.2f} us  sum={r[3]:8.2f} ms  max={r[4]:9.1f} us")
