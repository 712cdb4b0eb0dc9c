"""Generates pyaudiorestoration_amd/csrc/sinc_taps_gen.h: compile-time tap tables for the NT-specialised K_sinc loops.

On gfx950 a VALU instruction with an SGPR source issues at half the rate of one whose sources are VGPRs or a literal
(tools/ubench2.hip), so the specialised tap loops are fully unrolled and every table value becomes an instruction
literal.  Same forms and error budgets as choose_tap_modes() in sinc.hip (which serves every other NT at run time):

    R_n(q) = (win_n/pi)/(n^2 - q),  q = shift^2 in [0, 1/4],  win_n = float32(np.hanning(2 NT + 1))[NT + n]
    mode 0 (n < 5)          reciprocal:  R_n = 1/(a2_n + b_n q)
    mode 1 (5 <= n < p1)    2-term series A + B q + C q^2
    mode 2 (p1 <= n < p0)   linear minimax fit A + B q
    mode 3 (n >= p0)        constant (mid-range)

All coefficients carry the tap's sign (-1)^n.    usage: python tools/gen_sinc_taps.py [--check]"""
import math
import os
import sys

import numpy as np

CHUNK = 4
NTS = (32, 50)
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyaudiorestoration_amd", "csrc",
                   "sinc_taps_gen.h")


def tables(NT):
    n = np.arange(NT + 1)
    win = (0.5 + 0.5 * np.cos(np.pi * n / NT)).astype(np.float32).astype(np.float64)      # np.hanning(2NT+1)[NT+n] as f32
    win[NT] = 0.0
    K = win / np.pi
    n2 = (n * n).astype(np.float64)
    e0 = np.zeros(NT + 1)
    e1 = np.zeros(NT + 1)
    lin_a = np.zeros(NT + 1)
    lin_b = np.zeros(NT + 1)
    cst = np.zeros(NT + 1)
    for k in range(1, NT):
        if K[k] <= 0:
            continue
        f0, f1 = K[k] / n2[k], K[k] / (n2[k] - 0.25)
        cst[k] = 0.5 * (f0 + f1)
        e0[k] = 0.5 * (f1 - f0) * (2 * k + 1)
        slope = (f1 - f0) / 0.25
        qs = n2[k] - math.sqrt(K[k] / slope)
        gap = (f0 + slope * qs) - K[k] / (n2[k] - qs)
        lin_b[k] = slope
        lin_a[k] = f0 - 0.5 * gap
        e1[k] = 0.5 * gap * (2 * k + 1)

    def first_from(e, budget):
        frm = ((NT + CHUNK) // CHUNK) * CHUNK + 1
        tail = 0.0
        n0 = ((NT - 1) // CHUNK) * CHUNK + 1
        while n0 >= 5:
            tail += sum(e[k] for k in range(n0, min(n0 + CHUNK, NT)))
            if tail > budget:
                break
            frm = n0
            n0 -= CHUNK
        return frm
    p1 = first_from(e1, 3.0e-7)
    p0 = max(first_from(e0, 1.5e-6), p1)
    rows = []
    for k in range(NT + 1):
        sgn = -1.0 if k & 1 else 1.0
        if k == 0:
            rows.append((0, 0.0, -math.pi / win[0], 0.0))            # .B of row 0: b0 = -pi/win_0 (centre tap)
        elif k >= NT:
            rows.append((3, 0.0, 0.0, 0.0))                           # tap -NT: weight 0 (kept: 0 * NaN semantics)
        elif k < 5:
            rows.append((0, math.pi * n2[k] / win[k], -math.pi / win[k], 0.0))        # sign applied by the caller
        elif k < p1:
            A = win[k] / (math.pi * n2[k])
            rows.append((1, sgn * A, sgn * A / n2[k], sgn * A / (n2[k] * n2[k])))
        elif k < p0:
            rows.append((2, sgn * lin_a[k], sgn * lin_b[k], 0.0))
        else:
            rows.append((3, sgn * cst[k], 0.0, 0.0))
    return p1, p0, rows


def f32(x):
    v = float(np.float32(x))
    if v == 0.0:
        return "0.0f"
    return "%sf" % repr(v) if ("e" in repr(v) or "." in repr(v)) else "%s.0f" % repr(v)


# ---- Farrow bank of the fc = 1 path on the matrix cores (r03) -------------------------------------------------------
# The taps n >= 5 of the unity path are polynomials in q = shift^2 with FIXED coefficients: six fixed FIR filters on the
# input grid (e0 e1 e2: symmetric, coefficient c_k[|n|]; d0 d1 d2: antisymmetric, sign(n) |n| c_k[|n|]).  K_sinc evaluates
# them with v_mfma_f32_16x16x32_f16: D[(filter, position i)][block b] += A[(f, i)][k] B[k][b], B[k][b] = x16[8 b + k].
# This emits the constant A fragments: M tile rows m = (filter m >> 3, position i = m & 7); lane: row m = lane & 15,
# g = lane >> 4; element j of slice ks: tap n = 32 ks + 8 g + j - 31 - i.  float16 has 11 bits, so the dominant pair is
# split hi + lo * 2^-12; everything is scaled by 32 (keeps the smallest coefficients normal).
# Fragments 0-2 (e0 d0)h slices 0-2; 3-4 (e1 d1)h slices 0-1; 5-7 (e0 d0)lo slices 0-2; 8-9 (e2 d2)h slices 0-1.
FARROW_SCALE = 32.0
FARROW_LO = 4096.0
FARROW_FRAGS = ((0, 3, 0, 0), (0, 3, 1, 0), (0, 3, 2, 0), (1, 4, 0, 0), (1, 4, 1, 0),
                (0, 3, 0, 1), (0, 3, 1, 1), (0, 3, 2, 1), (2, 5, 0, 0), (2, 5, 1, 0))


# r06: the same bank for NT = 50 (the reference's default quality; block kernel only).  99 taps + 8 positions need FOUR 32-tap
# slices; (e1 d1) -- nonzero for |n| < 37 -- occupies slices 0-2, (e2 d2) -- |n| < 9 -- slices 1-2: 13 fragments,
# 0-3 (e0 d0)h, 4-6 (e1 d1)h slices 0-2, 7-10 (e0 d0)lo, 11-12 (e2 d2)h slices 1-2.  Scale 1024, not 32: the matrix cores flush
# float16 subnormals (found in r04), and at 32 the e0 coefficients of taps 47-49 (1e-7 .. 1.3e-6) would be lost.
FARROW50_SCALE = 1024.0
FARROW50_FRAGS = ((0, 3, 0, 0), (0, 3, 1, 0), (0, 3, 2, 0), (0, 3, 3, 0), (1, 4, 0, 0), (1, 4, 1, 0), (1, 4, 2, 0),
                  (0, 3, 0, 1), (0, 3, 1, 1), (0, 3, 2, 1), (0, 3, 3, 1), (2, 5, 1, 0), (2, 5, 2, 0))


def farrow_fragments(NT):
    """uint16 [frags][64][8]: the float16 bit patterns of the constant A fragments (NT = 32: 63 taps + 8 positions fit three
    32-tap slices, 10 fragments; NT = 50: four slices, 13 fragments)."""
    assert NT in (32, 50)
    if NT == 50:
        return farrow_fragments_general(50, FARROW50_FRAGS, FARROW50_SCALE)
    _, _, rows = tables(NT)

    def coef(f, n):
        a = abs(n)
        if a < 5 or a >= NT:
            return 0.0
        mode, A, B, C = rows[a]
        A, B, C = float(np.float32(A)), float(np.float32(B)), float(np.float32(C))
        sg = -1.0 if n < 0 else 1.0
        fa = float(np.float32(np.float32(a) * np.float32(A)))
        fb = float(np.float32(np.float32(a) * np.float32(B)))
        fc = float(np.float32(np.float32(a) * np.float32(C)))
        v = (A, B if mode <= 2 else 0.0, C if mode == 1 else 0.0, sg * fa, sg * fb if mode <= 2 else 0.0,
             sg * fc if mode == 1 else 0.0)[f]
        return FARROW_SCALE * v
    for f in (1, 2, 4, 5):                                 # (e1 d1) and (e2 d2) skip slice 2: no tap of theirs may sit there
        for i in range(8):
            for k in range(64, 96):
                assert coef(f, k - 31 - i) == 0.0
    for f in (0, 3):
        for i in range(8):
            assert coef(f, 96 - 31 - i) == 0.0 and coef(f, -32 - i) == 0.0   # everything of the window inside K = 96
    out = np.zeros((len(FARROW_FRAGS), 64, 8), dtype=np.float16)
    for fr, (fe, fd, ks, lo) in enumerate(FARROW_FRAGS):
        for lane in range(64):
            m, g = lane & 15, lane >> 4
            for j in range(8):
                cf = coef(fd if (m >> 3) else fe, 32 * ks + 8 * g + j - 31 - (m & 7))
                hi = np.float16(cf)
                out[fr, lane, j] = np.float16((cf - float(hi)) * FARROW_LO) if lo else hi
    return out.view(np.uint16)


def farrow_fragments_general(NT, frags, scale):
    _, _, rows = tables(NT)
    n_slices = (2 * NT - 1 + 8 + 31) // 32

    def coef(f, n):
        a = abs(n)
        if a < 5 or a >= NT:
            return 0.0
        mode, A, B, C = rows[a]
        A, B, C = float(np.float32(A)), float(np.float32(B)), float(np.float32(C))
        sg = -1.0 if n < 0 else 1.0
        fa = float(np.float32(np.float32(a) * np.float32(A)))
        fb = float(np.float32(np.float32(a) * np.float32(B)))
        fc = float(np.float32(np.float32(a) * np.float32(C)))
        v = (A, B if mode <= 2 else 0.0, C if mode == 1 else 0.0, sg * fa, sg * fb if mode <= 2 else 0.0,
             sg * fc if mode == 1 else 0.0)[f]
        return scale * v
    # every nonzero coefficient of a filter pair must sit in a slice the pair has a fragment for, and be a NORMAL float16
    have, flushed = {}, {}
    for fe, fd, ks, lo in frags:
        have.setdefault((fe, fd), set()).add(ks)
    for (fe, fd), sl in have.items():
        for f in (fe, fd):
            for i in range(8):
                for k in range(32 * n_slices):
                    c = coef(f, k - (NT - 1) - i)
                    assert c == 0.0 or (k // 32) in sl, (NT, f, i, k)
                    if f in (0, 3):                      # the constant terms: every coefficient a normal float16
                        assert c == 0.0 or abs(c) >= 2.0 ** -14, (NT, f, k, c)
                    elif c != 0.0 and abs(c) < 2.0 ** -14 and i == 0:
                        flushed[f] = flushed.get(f, 0.0) + abs(c) / scale * (0.25 if f in (1, 4) else 0.0625)
    # (higher-order coefficients of far taps that are float16 subnormals are zero to the matrix cores: what they would have added,
    # at the largest q, for a unit-peak signal)
    assert sum(flushed.values()) < 2.0e-7, flushed
    for f in (0, 3):
        for i in range(8):
            assert coef(f, 32 * n_slices - (NT - 1) - i) == 0.0 and coef(f, -NT - i) == 0.0
    out = np.zeros((len(frags), 64, 8), dtype=np.float16)
    for fr, (fe, fd, ks, lo) in enumerate(frags):
        for lane in range(64):
            m, g = lane & 15, lane >> 4
            for j in range(8):
                cf = coef(fd if (m >> 3) else fe, 32 * ks + 8 * g + j - (NT - 1) - (m & 7))
                hi = np.float16(cf)
                out[fr, lane, j] = np.float16((cf - float(hi)) * FARROW_LO) if lo else hi
    # (lo parts below float16's normal range are flushed by the matrix cores: they carry < 2^-26 of a coefficient)
    return out.view(np.uint16)


def render():
    out = ["// GENERATED by tools/gen_sinc_taps.py -- do not edit.  Compile-time tap tables of the NT-specialised K_sinc loops.",
           "#pragma once", "", "namespace par {", "",
           "template <int NT> struct TapTab;   // mode[n], A/B/C[n] (sign (-1)^n folded in; reciprocal rows: a2, b unsigned)", ""]
    for NT in NTS:
        p1, p0, rows = tables(NT)
        out.append("template <> struct TapTab<%d> {" % NT)
        out.append("  static constexpr int p1_from = %d, p0_from = %d;" % (p1, p0))
        out.append("  static constexpr int mode[%d] = {%s};" % (NT + 1, ", ".join(str(r[0]) for r in rows)))
        for name, idx in (("A", 1), ("B", 2), ("C", 3)):
            out.append("  static constexpr float %s[%d] = {%s};" % (name, NT + 1, ", ".join(f32(r[idx]) for r in rows)))
        out.append("};")
        out.append("")
    fr = farrow_fragments(32).reshape(-1, 2)
    words = (fr[:, 0].astype(np.uint32) | (fr[:, 1].astype(np.uint32) << 16))
    out.append("// Constant A fragments of the unity path's Farrow bank on the matrix cores (tools/gen_sinc_taps.py, farrow_fragments):")
    out.append("// [10 fragments][64 lanes][8 halves] as 32-bit words, 10 KB; scale %g, lo parts x %g." % (FARROW_SCALE, FARROW_LO))
    out.append("constexpr int kFarrowFrags = %d;" % len(FARROW_FRAGS))
    out.append("constexpr float kFarrowScaleInv = %sf, kFarrowLoInv = %sf;" % (repr(1.0 / FARROW_SCALE), repr(1.0 / FARROW_LO)))
    out.append("#ifdef __HIPCC__")
    out.append("__device__ const unsigned int kFarrowFrags32[%d] = {" % len(words))
    for i in range(0, len(words), 12):
        out.append("  " + ", ".join("0x%08xu" % w for w in words[i:i + 12]) + ",")
    out.append("};")
    out.append("#endif")
    out.append("")
    fr50 = farrow_fragments(50).reshape(-1, 2)
    words50 = (fr50[:, 0].astype(np.uint32) | (fr50[:, 1].astype(np.uint32) << 16))
    out.append("// ... and for NT = 50 (r06): [13 fragments][64 lanes][8 halves], four 32-tap slices; 0-3 (e0 d0)h, 4-6 (e1 d1)h slices 0-2,")
    out.append("// 7-10 (e0 d0)lo, 11-12 (e2 d2)h slices 1-2; scale %g, lo parts x %g." % (FARROW50_SCALE, FARROW_LO))
    out.append("constexpr int kFarrowFrags50 = %d;" % len(FARROW50_FRAGS))
    out.append("constexpr float kFarrowScaleInv50 = %sf;" % repr(1.0 / FARROW50_SCALE))
    out.append("#ifdef __HIPCC__")
    out.append("__device__ const unsigned int kFarrowFrags50_32[%d] = {" % len(words50))
    for i in range(0, len(words50), 12):
        out.append("  " + ", ".join("0x%08xu" % w for w in words50[i:i + 12]) + ",")
    out.append("};")
    out.append("#endif")
    out.append("")
    # ---- streaming kernel (sinc2.hip): bank over the taps 3 <= |n| <= 31, rows interleaved (e, d), + the (H, H1') pair
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import sinc2_model as M2
    fr2 = M2.bank_fragments().reshape(-1, 2)
    w2 = (fr2[:, 0].astype(np.uint32) | (fr2[:, 1].astype(np.uint32) << 16))
    out.append("// Constant A fragments of the streaming kernel's bank (tools/sinc2_model.py, bank_fragments): taps 3 <= |n| <= 31 as")
    out.append("// minimax polynomials in q (quadratic to n = %d, linear to 25, constant beyond), rows m = 2 i + (0: e, 1: d);" % max(n for n in M2.DEGS if M2.DEGS[n] == 2))
    out.append("// [13 fragments][64 lanes][8 halves]: 0-2 (e0 d0) hi, 3-4 (e1 d1) hi, 5-7 (e0 d0) lo x %g, 8-9 (e2 d2) hi, 10-12 (H H1'); e/d x %g." % (M2.LO, M2.SCALE))
    out.append("constexpr int kBank2Frags = %d;" % len(M2.FRAGS))
    out.append("constexpr float kBank2ScaleInv = %sf, kBank2LoInv = %sf;" % (repr(1.0 / M2.SCALE), repr(1.0 / M2.LO)))
    out.append("#ifdef PAR_WANT_BANK2")
    out.append("__device__ const unsigned int kBank2Frags32[%d] = {" % len(w2))
    for i in range(0, len(w2), 12):
        out.append("  " + ", ".join("0x%08xu" % w for w in w2[i:i + 12]) + ",")
    out.append("};")
    out.append("#endif")
    out.append("")
    fr3 = M2.moment_fragments().reshape(-1, 2)
    w3 = (fr3[:, 0].astype(np.uint32) | (fr3[:, 1].astype(np.uint32) << 16))
    out.append("// Constant A fragments of the moment filters m_i = sum_n (-1)^n win_n (n / 32)^i x[c + n], i = 0 .. 6, all taps |n| <= 31")
    out.append("// (tools/sinc2_model.py, moment_fragments; the fc < 1 correction of tools/sinc3_model.py): [15 fragments][64 lanes][8 halves]:")
    out.append("// 0-2 (m0 m1) hi, 3-5 (m0 m1) lo x %g, 6-8 (m2 m3), 9-11 (m4 m5), 12-14 (m6 -)." % M2.LO)
    out.append("constexpr int kBank3Frags = %d;" % len(M2.MOM_FRAGS))
    out.append("#ifdef PAR_WANT_BANK2")
    out.append("__device__ const unsigned int kBank3Frags32[%d] = {" % len(w3))
    for i in range(0, len(w3), 12):
        out.append("  " + ", ".join("0x%08xu" % w for w in w3[i:i + 12]) + ",")
    out.append("};")
    out.append("#endif")
    out.append("")
    out += ["}  // namespace par", ""]
    return "\n".join(out)


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    for NT in NTS:
        p1, p0, _ = tables(NT)
        print("NT", NT, "poly1 from", p1, "const from", p0)
