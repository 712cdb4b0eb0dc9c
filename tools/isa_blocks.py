"""Per-basic-block instruction census of one kernel in a hipcc -save-temps .s file.

usage: python tools/isa_blocks.py FILE.s MANGLED_SUBSTRING
Prints, for every basic block, the number of VALU (f32 / f64 / transcendental), SALU, LDS, SMEM, VMEM and
waitcnt instructions, and the branch at its end -- the table DESIGN.md's K_sinc instruction budget is built from."""
import re
import sys


def classify(op):
    if op.startswith("v_"):
        if op.startswith(("v_rcp", "v_sin", "v_cos", "v_sqrt", "v_rsq", "v_exp", "v_log")):
            return "trans"
        if "_f64" in op or op.endswith("_f64_e32") or "f64" in op:
            return "v64"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "vlane"
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and re.match(r"^_Z\w+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks = []
    cur = {"name": "entry", "line": start + 1, "n": {}, "br": []}
    for i in range(start + 1, end):
        l = lines[i].strip()
        if not l or l.startswith((";", ".p2align", ".loc", ".cfi", ".file", ".size", ".type", ".section")):
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            blocks.append(cur)
            cur = {"name": m.group(1), "line": i + 1, "n": {}, "br": []}
            continue
        if l.endswith(":") or l.startswith("."):
            continue
        op = l.split()[0]
        c = classify(op)
        cur["n"][c] = cur["n"].get(c, 0) + 1
        if c == "branch":
            cur["br"].append(" ".join(l.split()[:2]))
    blocks.append(cur)
    cols = ["valu", "v64", "trans", "vlane", "salu", "lds", "smem", "vmem", "wait", "barrier"]
    print(f"{'block':14s} {'line':>6s} " + " ".join(f"{c:>6s}" for c in cols) + "  branches")
    tot = {}
    for b in blocks:
        if not b["n"]:
            continue
        print(f"{b['name']:14s} {b['line']:6d} " + " ".join(f"{b['n'].get(c, 0):6d}" for c in cols) + "  " + "; ".join(b["br"]))
        for c in cols:
            tot[c] = tot.get(c, 0) + b["n"].get(c, 0)
    print(f"{'static total':14s} {'':6s} " + " ".join(f"{tot.get(c, 0):6d}" for c in cols))


if __name__ == "__main__":
    main()
