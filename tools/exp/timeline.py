"""Timeline of the pipelined step from a rocprofv3 kernel trace (rocpd sqlite): the K_sinc launches back to back, the gaps
between them, and which of the planners' kernels ran inside the gaps / beside K_sinc.
   rocprofv3 --kernel-trace -d out -o t -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline;  python tools/exp/timeline.py out/t_results.db"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
st, en = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
rows = c.execute(f"select name, {st}, {en} from kernels order by {st}").fetchall()
pipe = [(s, e) for n, s, e in rows if "k_sinc_pipe<1>" in n]
lst = [(s, e) for n, s, e in rows if "k_sinc_fused_list(" in n]
other = [(n, s, e) for n, s, e in rows if "k_sinc" not in n]
# the steady part: launches 35.. (behind the alone loops) of the pipelined steps
seq = pipe[-10:]
print(f"{len(pipe)} streaming launches; the last {len(seq)}:")
for k in range(1, len(seq)):
    s0, e0 = seq[k - 1]; s1, e1 = seq[k]
    le = max((e for s, e in lst if s >= e0 - 1000 and s < s1), default=e0)
    inside = [(n, s, e) for n, s, e in other if e > e0 and s < s1]
    gap_k = sum(min(e, s1) - max(s, le) for n, s, e in inside if min(e, s1) > max(s, le))
    beside = sum(min(e, e1) - max(s, s1) for n, s, e in other if min(e, e1) > max(s, s1))
    print(f"  pipe {(e0 - s0) / 1e3:8.1f} us | list ends +{(le - e0) / 1e3:6.1f} | next pipe starts +{(s1 - e0) / 1e3:6.1f} after the pipe's end "
          f"(other kernels' time inside that gap {gap_k / 1e3:6.1f} us, beside the next pipe {beside / 1e3:7.1f} us); period {(s1 - s0) / 1e3:8.1f}")
