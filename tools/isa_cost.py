"""Static issue-cost estimate of a loop in a gfx950 assembly listing (hipcc -S): sums, per instruction class, the SIMD cycles
measured by tools/ubench2.hip / tools/exp/issue_rate.hip (nominal 2.4 GHz cycles per wave64 instruction).
    python tools/isa_cost.py file.s <kernel-symbol-substring> <first-label> <last-label>
Counts the instructions between the two labels (inclusive of the blocks in between, in file order)."""
import re, sys
COST_PLAIN = 2.7
def cost(line):
    t = line.strip()
    op = t.split()[0]
    args = t[len(op):]
    sgpr_src = bool(re.search(r'(?<![\w\[])s\d+|s\[\d+:\d+\]|\bvcc\b|\bexec\b|\bm0\b', args.split(',', 1)[1] if ',' in args else ''))
    if op.startswith('v_mfma'): return 'mfma', 16.0
    if op.startswith('v_pk_'): return 'valu packed', 4.85
    if op.startswith(('v_rcp', 'v_sin', 'v_cos', 'v_rsq', 'v_sqrt', 'v_exp', 'v_log')): return 'valu transcendental', 8.8
    if op.startswith(('v_cmp', 'v_cmpx')): return 'valu compare', 4.4
    if op.startswith('v_cndmask'): return 'valu select', 4.4
    if op.startswith(('v_cvt', 'v_rndne', 'v_trunc', 'v_floor', 'v_fract')): return 'valu convert/round', 4.3
    if op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane')): return 'valu lane<->scalar', 4.4
    if op.startswith(('v_mul_lo', 'v_mul_hi', 'v_mul_u32', 'v_mul_i32', 'v_mad_u64', 'v_mad_i64')): return 'valu int mul', 4.8
    if op.startswith('v_') and ('_f64' in op or '_u64' in op or '_i64' in op or '_b64' in op): return 'valu 64-bit', 4.7
    if op.startswith('v_') and ('dpp' in t or 'sdwa' in t): return 'valu dpp/sdwa', 4.4
    if op.startswith('v_'):
        return ('valu plain (sgpr operand)', 4.5) if sgpr_src else ('valu plain', COST_PLAIN)
    if op.startswith('ds_'): return 'lds', 0.0
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem', 0.0
    if op.startswith('s_waitcnt'): return 'waitcnt', 0.0
    if op.startswith('s_nop'): return 's_nop', 0.0
    if op.startswith('s_'): return 'salu/branch', 0.0
    return 'other', 0.0
def main():
    path, kern, first, last = sys.argv[1:5]
    lines = open(path).read().split('\n')
    infn = False; on = False; tally = {}; n = 0
    for ln in lines:
        if re.match(r'^_Z\w+:', ln): infn = kern in ln
        if not infn: continue
        t = ln.strip()
        if t.startswith(first + ':'): on = True
        if not on: continue
        if t.endswith(':') and t.startswith(last + ':') and first != last: on = 'last'
        elif on == 'last' and t.endswith(':') and not t.startswith(last): break
        if not t or t.startswith((';', '.')) and not t.startswith('.LBB') or t.endswith(':'): continue
        k, c = cost(ln)
        d = tally.setdefault(k, [0, 0.0]); d[0] += 1; d[1] += c; n += 1
    tot = sum(v[1] for v in tally.values())
    for k, v in sorted(tally.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:28s} {v[0]:5d} instr  {v[1]:8.1f} cycles")
    print(f"  {'total':28s} {n:5d} instr  {tot:8.1f} vector-port cycles")
main()
