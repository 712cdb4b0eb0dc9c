#!/usr/bin/env python3
"""Turn the rocprofv3 sqlite outputs of tools/profile_round.sh into the committed text summaries:
     python tools/summarise_profiles.py gpurun_out/r01 r01
writes profiles/<tag>_kernel_stats.txt, profiles/<tag>_pmc.txt, profiles/<tag>_bench_default.json and
profiles/pmc_traffic.json (HBM bytes per output sample of K_sinc, used by bench.py's roofline.traffic)."""
import json
import os
import sqlite3
import subprocess
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)
py = sys.executable
stats = subprocess.check_output([py, os.path.join(root, "tools", "rocpd_stats.py"), os.path.join(src, "trace_results.db")], text=True)
open(os.path.join(prof, f"{tag}_kernel_stats.txt"), "w").write(
    "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline (default 3600-s workload)\n" + stats)


def counters(db, pat):
    c = sqlite3.connect(os.path.join(src, db))
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by 1, 2").fetchall()
    return {(k, n): (cnt, a) for k, n, cnt, a in rows if pat in k}


lines = ["# separate rocprofv3 --kernel-trace --pmc passes over: python bench.py --steps 2 --warmup 1 --no-cpu-baseline",
         "# per-launch averages; FETCH_SIZE/WRITE_SIZE in KiB (FETCH_SIZE under-reports coalesced reads 2x on gfx950)"]
vals = {}
for db in ("fetch", "write", "sq", "lds", "grbm"):
    path = os.path.join(src, db + "_results.db")
    if not os.path.exists(path):
        continue
    for (k, n), (cnt, a) in sorted(counters(db + "_results.db", "par::").items()):
        if any(s in k for s in ("k_sinc", "k_pos_fill", "k_seg_sum", "k_stft")):
            lines.append(f"{k[:60]:60s} {n:24s} n={cnt:3d} avg={a:18.1f}")
            short = "k_sinc" if "k_sinc" in k else ("k_pos_fill" if "k_pos_fill" in k else ("k_seg_sum" if "k_seg_sum" in k else "k_stft"))
            if short == "k_sinc" and ", 2>" in k:
                short = "k_sinc_stereo"                       # the config-5 secondary line, not the timed workload
            vals[(short, n)] = a
open(os.path.join(prof, f"{tag}_pmc.txt"), "w").write("\n".join(lines) + "\n")
bench = json.loads(open(os.path.join(src, "bench_default.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(prof, f"{tag}_bench_default.json"), "w"), indent=1)
fetch = vals.get(("k_sinc", "FETCH_SIZE"))
write = vals.get(("k_sinc", "WRITE_SIZE"))
if fetch and write:
    n = bench["roofline"]["samples_per_launch"]
    t = {"kernel": bench["roofline"].get("kernel", "k_sinc") + (" (fused: positions regenerated in LDS)" if "fused" in bench["config"]["step"].split(";")[0] else ""),
         "samples_per_launch": n, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
         "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "hbm_bytes_per_sample": (2 * fetch + write) * 1024 / n,
         "correction": "FETCH_SIZE x2 (gfx950 coalesced-read under-count, MI355X_MICROARCH.md HBM section), KiB units",
         "source": f"profiles/{tag}_pmc.txt"}
    json.dump(t, open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
    print(t)
print(stats[:1500])
