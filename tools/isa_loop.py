"""Instruction census of the loops that hold a marker instruction (default: the pipelined K_sinc loops' `s_waitcnt vmcnt(5)`) in a hipcc -S
listing: for each marker, the innermost loop around it (header label .. the backward branch to it), priced with tools/isa_cost.py's table.
    python tools/isa_loop.py file.s <kernel-symbol-substring> [marker-regex]"""
import re, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "isa_cost.py")).read().split("def main")[0]
exec(src)
path, kern = sys.argv[1:3]
marker = re.compile(sys.argv[3] if len(sys.argv) > 3 else r"s_waitcnt vmcnt\(5\)")
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and kern in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
for mi, l in enumerate(body):
    if not marker.search(l): continue
    # innermost loop: the closest label above the marker that some later branch jumps back to
    best = None
    for lab, li in labels.items():
        if li > mi: continue
        for bi in range(mi, len(body)):
            if re.search(r"s_c?branch\w*\s+" + re.escape(lab) + r"\b", body[bi]):
                if best is None or li > best[0]: best = (li, bi, lab)
                break
    if not best: continue
    li, bi, lab = best
    tally, n = {}, 0
    for t in body[li:bi + 1]:
        u = t.strip()
        if not u or u.startswith(";") or (u.startswith(".") and not u.startswith(".LBB")) or u.endswith(":"): continue
        k, c = cost(t); d = tally.setdefault(k, [0, 0.0]); d[0] += 1; d[1] += c; n += 1
    tot = sum(v[1] for v in tally.values())
    print(f"loop {lab} (lines {li}..{bi}): {n} instructions, {tot:.0f} vector-port cycles; " + ", ".join(f"{k} {v[0]}" for k, v in sorted(tally.items(), key=lambda kv: -kv[1][0])))
