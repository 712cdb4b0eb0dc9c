#!/usr/bin/env python3
"""Per-kernel PMC counter averages from a rocprofv3 rocpd sqlite file:
   python tools/rocpd_pmc.py file.db [kernel-substring]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
q = "select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection group by kernel_name, counter_name"
try:
    rows = c.execute(q).fetchall()
except Exception as e:
    print("columns:", cols)
    raise
for k, n, cnt, a, s in rows:
    if pat in k:
        print(f"{k[:48]:48s} {n:28s} n={cnt:4d} avg={a:16.1f}")
