"""Timeline of the pipelined step's mono launches (r06: k_sinc_pipe<1, 2>, k_sinc_pipe<1, 1>, k_sinc_fused_list, the memset in front)
from a rocprofv3 kernel trace: per file, the four kernels' durations and the gaps between them.
   rocprofv3 --kernel-trace -d out -o t -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-parity;  python tools/exp/timeline2.py out/t_results.db"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = "kernels" if "kernels" in tabs else next(t for t in tabs if "kernel" in t.lower())
cols = [r[1] for r in c.execute(f"pragma table_info({kt})")]
st, en = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
nm = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select {nm}, {st}, {en} from {kt} order by {st}").fetchall()
k2 = [(s, e) for n, s, e in rows if "k_sinc_pipe<1, 2>" in n]
seq = k2[-9:]
tot = []
for (s0, e0), (s1, e1) in zip(seq[:-1], seq[1:]):
    inside = [(n, s, e) for n, s, e in rows if s >= s0 and s < s1]
    k1 = next(((s, e) for n, s, e in inside if "k_sinc_pipe<1, 1>" in n), None)
    ls = next(((s, e) for n, s, e in inside if "k_sinc_fused_list(" in n), None)
    fill = [(n, s, e) for n, s, e in inside if "fillBuffer" in n and s > (ls[1] if ls else e0)]
    others = sum(e - s for n, s, e in inside if "k_sinc" not in n)
    print(f"<1,2> {(e0 - s0) / 1e3:7.1f} | gap {(k1[0] - e0) / 1e3:6.1f} | <1,1> {(k1[1] - k1[0]) / 1e3:7.1f} | gap {(ls[0] - k1[1]) / 1e3:6.1f} | list {(ls[1] - ls[0]) / 1e3:5.1f} "
          f"| to the next <1,2> {(s1 - ls[1]) / 1e3:6.1f} (fills there: {len(fill)}) | period {(s1 - s0) / 1e3:7.1f} | other kernels' time in the period {others / 1e3:7.1f}")
    tot.append((s1 - s0) / 1e3)
print("mean period", sum(tot) / len(tot))
