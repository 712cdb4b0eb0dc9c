#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace as the --stats table:
   python tools/rocpd_stats.py gpurun_out/prof/x_results.db [> profiles/...txt]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                 "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                 "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print(f"{'kernel':70s} {'calls':>6s} {'total_us':>11s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>7s} {'grid':>10s} {'wg':>5s}")
for n, cnt, s, a, mn, mx, vg, sg, lds, gx, wx in rows:
    print(f"{n[:70]:70s} {cnt:6d} {s/1e3:11.1f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f} {vg:5d} {sg:5d} {lds:7d} {gx:10d} {wx:5d}")
