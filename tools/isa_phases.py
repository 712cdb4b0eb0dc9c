#!/usr/bin/env python3
"""Static instruction census of one kernel BY SOURCE FUNCTION (after inlining), from a hipcc -save-temps -gline-tables-only
.s file: every instruction is attributed to the function of csrc/<file> its .loc line lies in.
    python tools/isa_phases.py sinc.hip k_sinc_fusedILi1ELi32ELi4 [extra hipcc flags]
Instructions inlined from compiler headers (fmaf ...) are charged to the last csrc function seen before them.
Everything in the NT-specialised kernels is unrolled, so static counts per function = executed counts per wave on that
path (unity and general taps are different functions; only one of them runs for a given wave)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pyaudiorestoration_amd import build as B
from isa_blocks import classify


def functions_of(path):
    """[(first_line, name)] of the function definitions in a source file (brace at column 0 closes them)."""
    out, lines = [], open(path).read().split("\n")
    for i, l in enumerate(lines):
        m = re.match(r"^(?:template\s*<[^>]*>\s*)?(?:static\s+|__device__\s+|__global__\s+|__host__\s+|__forceinline__\s+|__noinline__\s+|"
                     r"inline\s+|constexpr\s+|__launch_bounds__\([^)]*\)\s+)*[\w:<>,\s\*&]+?\b(\w+)\s*\([^;]*$", l)
        if m and not l.startswith((" ", "\t", "//", "#", "}")) and m.group(1) not in ("if", "for", "while", "switch", "return"):
            out.append((i + 1, m.group(1)))
    return out


def main():
    src, key, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    tmp = tempfile.mkdtemp()
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + B.PER_FILE.get(src, []) + extra +
                          ["-gline-tables-only", "-save-temps", "-c", os.path.join(B.CSRC, src), "-o", os.path.join(tmp, "o.o")], cwd=tmp)
    s = next(os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith("gfx950.s"))
    lines = open(s).read().split("\n")
    files = {}
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    fn_tabs = {}
    cur = ("?", 0)
    last_src = "?"
    table = {}
    for i in range(start + 1, end):
        l = lines[i].strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        if not l or l.startswith((";", ".", "//")) or l.endswith(":"):
            continue
        op = l.split()[0]
        f, ln = cur
        if f not in fn_tabs:
            p = os.path.join(B.CSRC, f)
            fn_tabs[f] = functions_of(p) if os.path.exists(p) else []
        name = f
        for first, n in fn_tabs[f]:
            if first <= ln:
                name = f"{f}:{n}"
        if not fn_tabs[f]:                         # a compiler header (fmaf, shuffles): charge the csrc function it was inlined into
            name = last_src                        # -- approximated by the last csrc function seen (same unrolled block)
        else:
            last_src = name
        c = classify(op)
        table.setdefault(name, {}).setdefault(c, 0)
        table[name][c] += 1
    cols = ["valu", "v64", "trans", "vlane", "salu", "lds", "smem", "vmem", "wait", "branch"]
    print(f"{'source function':44s} " + " ".join(f"{c:>6s}" for c in cols))
    tot = {}
    for name, n in sorted(table.items(), key=lambda kv: -sum(kv[1].values())):
        print(f"{name:44s} " + " ".join(f"{n.get(c, 0):6d}" for c in cols))
        for c in cols:
            tot[c] = tot.get(c, 0) + n.get(c, 0)
    print(f"{'total':44s} " + " ".join(f"{tot.get(c, 0):6d}" for c in cols))


if __name__ == "__main__":
    main()
