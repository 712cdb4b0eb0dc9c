#!/usr/bin/env python3
"""Per-stage instruction census of the streaming kernel's two hot loops (fc = 1 and fc < 1), from the compiler's listing.
    python tools/isa_census.py [NCH [KIND]] [extra hipcc flags]   -> prints the table committed as profiles/r06_isa_census.txt
Every instruction between a loop's `s_waitcnt vmcnt(N)` head and its back edge is attributed to the STAGE its .loc source
line belongs to (source ranges of sinc2.hip named below; inlined header code is charged to the last csrc line seen)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyaudiorestoration_amd import build as B

def stage_table(src):
    """(first, last, stage) source ranges, found by marker comments / function names so that edits do not break them"""
    lines = open(src).read().split("\n")
    def find(pat, start=0):
        for i in range(start, len(lines)):
            if re.search(pat, lines[i]):
                return i + 1
        raise KeyError(pat)
    def end_of(first):            # a column-0 function: closes at the first "^}" behind it
        for i in range(first, len(lines)):
            if lines[i].startswith("}"):
                return i + 1
    T = []
    for name, pat in (("place_row", r"S2Row s2_place_row"), ("bank", r"void bank_image3m"), ("out_row", r"float s3_out_row"),
                      ("convert", r"bool s3_convert\("), ("convert", r"bool s3_convert_ch\("), ("out_row", r"float sinpi_poly")):
        a = find(r"__device__ __forceinline__ " + pat) if "S2Row" not in pat else find(pat)
        T.append((a - 1, end_of(a), name))
    a = find(r"auto place = \[&\]"); b = find(r"^\s*S3Pass P;", a)
    T.append((a, b, "place"))
    a = find(r"auto place_next = \[&\]"); b = find(r"^\s*};", a)
    T.append((a, b, "place_next"))
    a = find(r"auto out_pass = \[&\]"); b = find(r"auto store_pass = ", a)
    T.append((a, b - 1, "out_pass"))
    a = find(r"auto store_pass = \[&\]"); b = find(r"auto convert_chunk", a)
    T.append((a, b - 1, "store"))
    a = find(r"auto fetch_records = \[&\]"); b = find(r"auto push_tile", a)
    T.append((a, b - 1, "fetch"))
    for name, pat in (("dma", r"void dma_dword\("), ("dma", r"void dma_dwordx4"), ("dma", r"void dma_chunk128"), ("dma", r"void dma_chunk256"),
                      ("fence", r"void wave_lds_fence")):
        a = find(pat); T.append((a, end_of(a), name))
    return T

CLASS = (("mfma", r"v_mfma"), ("trans", r"v_(rcp|rsq|sqrt|sin|cos|exp|log)_"), ("valu", r"v_"), ("lds", r"ds_"),
         ("vmem", r"(global|buffer|flat|scratch)_"), ("salu", r"s_"))

SLOW = r"v_(cvt|rndne|trunc|floor|fract|med3|bfe|lshl_or|lshl_add|add3|and_or|or3|xad|bitop3|perm|cmp|cndmask|readlane|readfirstlane|writelane|permlane|alignbit|mad_|mul_lo|mul_hi|sad|min3|max3|bfi|lshrrev_b64|lshlrev_b64)"
def port_cycles(l):
    """vector-port price of one instruction (nominal-clock cycles per wave64 instruction and SIMD, tools/exp/valu_forms.hip, r06): plain
    float / integer VOP2-style forms 2.4 (a literal or inline constant costs nothing), anything with an SGPR operand and the
    conversions, selects, compares, three-operand integer forms 4.3, transcendentals 8.2, packed float 4.8, MFMA 16x16x32 16"""
    op = l.split()[0]
    args = l[len(op):]
    src = args.split(",", 1)[1] if "," in args else ""
    if op.startswith("v_mfma"): return 16.0
    if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)_", op): return 8.2
    if op.startswith("v_pk_"): return 4.8
    if not op.startswith("v_"): return 0.0
    if re.match(SLOW, op) or "_f64" in op or "_u64" in op or "_b64" in op: return 4.3
    if re.search(r"(?<![\w\[])s\d+|s\[\d+:\d+\]|\bvcc\b|\bexec\b|\bm0\b", src): return 4.3
    return 2.4

def main():
    nch = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].isdigit() else "1"
    kind = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2].isdigit() else ("2" if nch == "1" else "0")
    dump = next((a[7:] for a in sys.argv[1:] if a.startswith("--dump=")), None)     # --dump=STAGE: list that stage's instructions
    extra = [a for a in sys.argv[1:] if not a.isdigit() and not a.startswith("--dump=")]
    tmp = tempfile.mkdtemp()
    src = os.path.join(B.CSRC, "sinc2.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + B.PER_FILE.get("sinc2.hip", []) + extra +
                          ["-gline-tables-only", "-save-temps", "-c", src, "-o", os.path.join(tmp, "o.o")], cwd=tmp,
                          stderr=subprocess.DEVNULL)
    s = next(os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith("gfx950.s") and f.startswith("sinc2"))
    L = open(s).read().split("\n")
    files = {}
    for l in L:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m: files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    start = next(i for i, l in enumerate(L) if re.match(r"^_ZN3par11k_sinc_pipeILi%sELi%sEEE" % (nch, kind), l))
    end = next(i for i in range(start, len(L)) if L[i].startswith(".Lfunc_end"))
    for i in range(end, min(end + 60, len(L))):
        if re.search(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy|NumSgprs|LDSByteSize)", L[i]): print(L[i].strip())
    stages = stage_table(src)
    heads = [i for i in range(start, end) if re.search(r"s_waitcnt vmcnt\((5|7)\)", L[i])]
    for hi, h in enumerate(heads):
        # loop header: the nearest "Parent Loop" label above the head; body: every block that names it as its header
        hl = next(i for i in range(h, start, -1) if re.match(r"^\.LBB\d+_\d+:.*Parent Loop", L[i]))
        lab = L[hl].split(":")[0].lstrip(".L")
        # the latch is the block in front of the header (it falls through into it); the body ends at the last branch to the latch
        latch = next(L[i].split(":")[0] for i in range(hl - 1, start, -1) if re.match(r"^\.LBB\d+_\d+:", L[i]))
        back = max(i for i in range(hl, end) if re.search(r"s_c?branch\w*\s+%s\b" % re.escape(latch), L[i]))
        h = hl
        cur, last_stage = ("?", 0), "loop"
        tab = {}
        for i in range(h, back + 1):
            l = L[i].strip()
            m = re.match(r"\.loc\s+(\d+)\s+(\d+)", l)
            if m: cur = (files.get(int(m.group(1)), "?"), int(m.group(2))); continue
            if not l or l.startswith((";", ".", "//")) or l.endswith(":"): continue
            op = l.split()[0]
            if cur[0] == "sinc2.hip":
                st = next((n for a, b, n in stages if a <= cur[1] <= b), "loop")
                last_stage = st
            else:
                st = last_stage
            cls = next(c for c, p in CLASS if re.match(p, op))
            tab.setdefault(st, {}).setdefault(cls, 0)
            tab[st][cls] += 1
            tab[st]["port"] = tab[st].get("port", 0.0) + port_cycles(l)
            if dump == st and cls in ("valu", "trans", "mfma"):
                print(f"   [{hi}] {cur[1]:5d} {port_cycles(l):4.1f}  {l}")
        n_mfma = sum(v.get("mfma", 0) for v in tab.values())
        what = "fc = 1" if n_mfma <= 16 else ("fc < 1, moment correction to order 5 (1 - fc <= 0.0105)" if n_mfma <= 29 else "fc < 1, moment correction to order 6")
        print(f"\nloop {hi} ({what}): listing lines {h}..{back}  [static; the placement's "
              "two variants (with / without second pieces) are both counted, one runs]")
        cols = ["valu", "trans", "mfma", "lds", "vmem", "salu"]
        print(f"{'stage':12s}" + "".join(f"{c:>7s}" for c in cols) + "  port cycles")
        tot = dict.fromkeys(cols + ["port"], 0)
        for st in sorted(tab, key=lambda k: -tab[k].get("port", 0.0)):
            print(f"{st:12s}" + "".join(f"{tab[st].get(c, 0):7d}" for c in cols) + f"  {tab[st].get('port', 0.0):8.0f}")
            for c in cols + ["port"]: tot[c] += tab[st].get(c, 0)
        print(f"{'total':12s}" + "".join(f"{tot[c]:7d}" for c in cols) + f"  {tot['port']:8.0f}   = {tot['port'] / 128:.2f} SIMD cycles per output")

if __name__ == "__main__":
    main()
