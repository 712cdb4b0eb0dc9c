#!/usr/bin/env python3
"""Turn the rocprofv3 sqlite outputs of tools/profile_round.sh into the committed text summaries:
     python tools/summarise_profiles.py gpurun_out/r01 r01
writes profiles/<tag>_kernel_stats.txt, profiles/<tag>_pmc.txt, profiles/<tag>_bench_default.json and
profiles/pmc_traffic.json (HBM bytes per output sample of K_sinc, used by bench.py's roofline.traffic) and
profiles/pmc_valu.json (VALU lane-instructions per output sample of K_sinc, used by bench.py's roofline_valu)."""
import json
import os
import sqlite3
import subprocess
import sys


def pick_counters(rows, symbols):
    """Counters of ONE timed K_sinc launch.  rows: {(kernel_name, counter_name): per-launch average} as rocprofv3 names kernels
    (e.g. 'void par::k_sinc_pipe<1>(par::S2Args)'); symbols: the template-qualified names the bench line says a timed
    launch consists of (roofline.kernel_symbols, e.g. ['k_sinc_pipe<1>', 'k_sinc_fused_list']).  A kernel belongs to a
    symbol when 'par::<symbol>(' occurs in its name -- 'k_sinc_fused<1, 32, 4>' does not pick up 'k_sinc_fused<2, 32, 4>' or
    'k_sinc_fused_list', and a second K_sinc kernel the same run happens to launch (r04: the opt-in moment kernel, whose rows
    overwrote the timed kernel's in profiles/pmc_*.json) is simply not asked for.  Returns ({counter: sum over the symbols},
    [missing symbols])."""
    out, missing = {}, []
    for sym in symbols:
        hit = {n: a for (k, n), a in rows.items() if ("par::" + sym + "(") in k}
        if not hit:
            missing.append(sym)
        for n, a in hit.items():
            out[n] = out.get(n, 0.0) + a
    return out, missing


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    os.makedirs(prof, exist_ok=True)
    py = sys.executable
    sys.path.insert(0, root)
    from pyaudiorestoration_amd import build as _build
    digest = _build.source_digest()            # the box has no .git: the kernel sources themselves are the version
    stats = subprocess.check_output([py, os.path.join(root, "tools", "rocpd_stats.py"), os.path.join(src, "trace_results.db")], text=True)
    open(os.path.join(prof, f"{tag}_kernel_stats.txt"), "w").write(
        "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline (default 3600-s workload)\n" + stats)

    def counters(db, pat):
        c = sqlite3.connect(os.path.join(src, db))
        rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by 1, 2").fetchall()
        return {(k, n): (cnt, a) for k, n, cnt, a in rows if pat in k}

    lines = ["# separate rocprofv3 --kernel-trace --pmc passes over: python bench.py --steps 2 --warmup 1 --no-cpu-baseline",
             "# per-launch averages; FETCH_SIZE/WRITE_SIZE in KiB (FETCH_SIZE under-reports coalesced reads 2x on gfx950)"]
    rows = {}
    for db in ("fetch", "write", "sq", "lds", "grbm"):
        path = os.path.join(src, db + "_results.db")
        if not os.path.exists(path):
            continue
        for (k, n), (cnt, a) in sorted(counters(db + "_results.db", "par::").items()):
            names = ("k_sinc", "k_pos_fill", "k_seg_sum", "k_block_rec", "k_tile_seg", "k_offs", "k_off_", "k_scan", "k_len_", "k_trim", "k_stft", "k_istft")
            if any(s_ in k for s_ in names):
                lines.append(f"{k[:72]:72s} {n:24s} n={cnt:3d} avg={a:18.1f}")
                rows[(k, n)] = a
    open(os.path.join(prof, f"{tag}_pmc.txt"), "w").write("\n".join(lines) + "\n")
    bench = json.loads(open(os.path.join(src, "bench_default.json")).read().strip().splitlines()[-1])
    json.dump(bench, open(os.path.join(prof, f"{tag}_bench_default.json"), "w"), indent=1)
    symbols = bench["roofline"]["kernel_symbols"]            # what ONE timed K_sinc launch consists of
    vals, missing = pick_counters(rows, symbols)
    if missing:
        print("no counters for", missing, "-- pmc_traffic.json / pmc_valu.json left as they are")
        vals = {}
    n = bench["roofline"]["samples_per_launch"]
    fetch, write = vals.get("FETCH_SIZE"), vals.get("WRITE_SIZE")
    if fetch and write:
        t = {"kernel": bench["roofline"].get("kernel", "k_sinc"), "kernel_symbols": symbols, "fused": "fused" in bench["config"]["step"].split(";")[0],
             "samples_per_launch": n, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
             "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "hbm_bytes_per_sample": (2 * fetch + write) * 1024 / n,
             "correction": "FETCH_SIZE x2 (gfx950 coalesced-read under-count, MI355X_MICROARCH.md HBM section), KiB units; summed over kernel_symbols",
             "source": f"profiles/{tag}_pmc.txt", "source_digest": digest}
        json.dump(t, open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
        print(t)
    valu = vals.get("SQ_INSTS_VALU")
    if valu:
        fma = 58.76
        try:                                                      # tools/ubench.hip's measured v_fma_f32 stream, if this round re-ran it
            for ln in open(os.path.join(prof, f"{tag}_ubench_gfx950.txt")):
                if ln.startswith("fma_stream_Tlaneops"):
                    fma = float(ln.split()[1])
        except OSError:
            pass
        v = {"kernel": bench["roofline"].get("kernel", "k_sinc"), "kernel_symbols": symbols, "samples_per_launch": n,
             "SQ_INSTS_VALU_per_launch": valu, "valu_lane_instr_per_output": valu * 64 / n, "fma_stream_Tlaneops": fma,
             "SQ_INSTS_MFMA_note": "SQ_INSTS_VALU counts wave64 instructions (matrix-core instructions included); x64 lanes / output samples of the launch, summed over kernel_symbols",
             "source": f"profiles/{tag}_pmc.txt", "source_digest": digest}
        json.dump(v, open(os.path.join(prof, "pmc_valu.json"), "w"), indent=1)
        print(v)
    print(stats[:1500])


if __name__ == "__main__":
    main()
