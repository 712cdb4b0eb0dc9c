"""Numerical model of K_sinc v2 (csrc/sinc2.hip): float64 reference against the formulation the kernel evaluates, with the
float16 hi/lo operand split of the matrix-core bank emulated.  Also the generator of the bank's constant A fragments
(tools/gen_sinc_taps.py imports `bank_fragments`).

  window:  out = sum_{k=-NT}^{NT-1} x[c+k] win_k sin(pi fc (k-s)) / (pi (k-s)),  c = rint(p), s = p - c, fc = min(1, 1/period)
  near taps |k| <= 2 on the vector units with the lane's own fc;
  far taps 3 <= |k| <= 31 as a Farrow bank in q = s^2 on the matrix cores (tap -32 has window weight 0):
     R_n(q) = (win_n/pi)/(n^2-q) ~ A_n + B_n q + C_n q^2   (minimax quadratics near the centre, linear / constant further out)
     e_i = sum_n (-1)^n X_n(i) (x[c+n] + x[c-n]),  d_i = sum_n (-1)^n n X_n(i) (x[c+n] - x[c-n]),  X(0,1,2) = A,B,C
     U(x) = s e(q) + d(q)         fc = 1:  far = -sin(pi s) U(x)
  fc < 1 (pass-uniform g0 = 1 - fc0, image phase k = window index, K = window index of the centre):
     A_k = x_k sin(pi g0 k), B_k = x_k cos(pi g0 k), psi = pi (s - g0 (K + s)), eps = fc - fc0
     far = -cos(psi) U(A) - sin(psi) U(B) + eps (cos(psi) H(B) - sin(psi) H(A))
           + (pi eps^2 / 2) (cos(psi) H1(A) + sin(psi) H1(B)),   H(y) = sum (-1)^n win_n y[c+n],  H1(y) = sum (-1)^n win_n (n - s) y[c+n]
usage: python tools/sinc2_model.py        (prints error tables; asserts the budgets the kernel relies on)"""
import math
import numpy as np

NT = 32
NEAR = 2                    # taps |n| <= NEAR stay on the vector units
SCALE = 1024.0              # bank coefficients are stored x 1024: the smallest one (tap 31) stays a NORMAL float16 -- the matrix
                            # cores flush subnormal float16 operands to zero (measured r04: sparse 2e-5 errors where an image crossed 0)
LO = 4096.0                 # lo parts are stored x 4096
n_all = np.arange(-NT, NT + 1)
WIN = np.hanning(2 * NT + 1).astype(np.float32).astype(np.float64)     # win[k + NT]


def win_n(n):
    return WIN[NT + abs(int(n))]


def minimax_poly(f, deg, lo=0.0, hi=0.25, iters=12):
    """Remez on [lo, hi] for a smooth f: returns coefficients c[0..deg] (ascending) and the max error."""
    m = deg + 2
    x = 0.5 * (lo + hi) + 0.5 * (hi - lo) * np.cos(np.pi * np.arange(m) / (m - 1))[::-1]
    grid = np.linspace(lo, hi, 4001)
    for _ in range(iters):
        V = np.vander(x, deg + 1, increasing=True)
        Aeq = np.hstack([V, ((-1.0) ** np.arange(m))[:, None]])
        sol = np.linalg.solve(Aeq, f(x))
        c = sol[:-1]
        err = np.polyval(c[::-1], grid) - f(grid)
        # new extrema: local maxima of |err| between sign changes
        idx = [0]
        for i in range(1, len(grid) - 1):
            if (err[i] - err[i - 1]) * (err[i + 1] - err[i]) <= 0:
                idx.append(i)
        idx.append(len(grid) - 1)
        # pick m alternating extrema with largest |err|
        ext = []
        for i in idx:
            if ext and np.sign(err[i]) == np.sign(err[ext[-1]]):
                if abs(err[i]) > abs(err[ext[-1]]):
                    ext[-1] = i
            else:
                ext.append(i)
        while len(ext) > m:
            if abs(err[ext[0]]) < abs(err[ext[-1]]):
                ext.pop(0)
            else:
                ext.pop()
        if len(ext) < m:
            break
        x = grid[ext]
    return c, float(np.max(np.abs(err)))


def bank_table():
    """rows[n] = (A, B, C) of R_n(q) for 3 <= n <= 31, the polynomial degree chosen per tap from worst-case budgets: a tap's
    approximation error times the largest bracket n |D| + |s| |E| <= 2 n + 1 (unit-peak input), summed over the taps of a
    form.  Constant fits from n = 26 (the (e1 d1) and (e2 d2) filters must fit two 32-tap slices: |n| <= 25), linear fits
    from the first n whose tail of linear-fit errors stays below 1.5e-7, minimax quadratics inside."""
    fits = {}
    for n in range(NEAR + 1, NT):
        K = win_n(n) / math.pi
        f = lambda q, K=K, n=n: K / (n * n - q)
        fits[n] = [minimax_poly(f, deg) for deg in (0, 1, 2)]
    p0 = 26
    tail, p1 = 0.0, p0
    for n in range(p0 - 1, NEAR, -1):
        tail += fits[n][1][1] * (2 * n + 1)
        if tail > 1.5e-7:
            break
        p1 = n
    rows, degs, used = {}, {}, [0.0, 0.0, 0.0]
    for n in range(NEAR + 1, NT):
        deg = 0 if n >= p0 else (1 if n >= p1 else 2)
        c, e = fits[n][deg]
        rows[n] = tuple(list(c) + [0.0] * (2 - deg))
        degs[n] = deg
        used[deg] += e * (2 * n + 1)
    return rows, degs, used


ROWS, DEGS, BUDGET = bank_table()


def coef(f, n):
    """coefficient of tap n (-31..31) of filter f: 0 1 2 = e0 e1 e2 (symmetric), 3 4 5 = d0 d1 d2 (antisymmetric, x n),
    6 = H (window with alternating sign), 7 = H1' = n H.  Sign (-1)^n folded in; e/d scaled by SCALE."""
    a = abs(n)
    if a <= NEAR or a >= NT:
        return 0.0
    sg = -1.0 if (a & 1) else 1.0
    if f < 3:
        return SCALE * sg * ROWS[a][f]
    if f < 6:
        return SCALE * sg * n * ROWS[a][f - 3]
    if f == 6:
        return sg * win_n(a)
    return sg * win_n(a) * n


# ---- constant A fragments (v_mfma_f32_16x16x32_f16): row m = 2 i + f', lane: m = lane & 15, g = lane >> 4; element j of slice ks
# is tap n = 32 ks + 8 g + j - 31 - i.  A lane of the result then holds rows 4 g' .. 4 g' + 3 = (e, d) of positions 2 g', 2 g' + 1.
# Fragments: 0-2 (e0 d0) hi slices 0-2; 3-4 (e1 d1) hi slices 0-1; 5-7 (e0 d0) lo slices 0-2; 8-9 (e2 d2) hi slices 0-1;
# 10-12 (H H1') hi slices 0-2.
FRAGS = ((0, 3, 0, 0), (0, 3, 1, 0), (0, 3, 2, 0), (1, 4, 0, 0), (1, 4, 1, 0),
         (0, 3, 0, 1), (0, 3, 1, 1), (0, 3, 2, 1), (2, 5, 0, 0), (2, 5, 1, 0),
         (6, 7, 0, 0), (6, 7, 1, 0), (6, 7, 2, 0))


def bank_fragments():
    for f in (1, 2, 4, 5):                      # (e1 d1), (e2 d2) must not need slice 2
        for i in range(8):
            for k in range(64, 96):
                assert coef(f, k - 31 - i) == 0.0, (f, i, k)
    out = np.zeros((len(FRAGS), 64, 8), dtype=np.float16)
    for fr, (fe, fd, ks, lo) in enumerate(FRAGS):
        for lane in range(64):
            m, g = lane & 15, lane >> 4
            i, fsel = m >> 1, m & 1
            for j in range(8):
                n = 32 * ks + 8 * g + j - 31 - i
                cf = coef(fd if fsel else fe, n) if -NT < n < NT else 0.0
                hi = np.float16(cf if abs(cf) >= 2.0 ** -14 else 0.0)
                assert cf == 0.0 or abs(cf) >= 2.0 ** -14, (fr, lane, j, cf)     # no coefficient is lost to the flush
                out[fr, lane, j] = np.float16((cf - float(hi)) * LO) if lo else hi
    return out.view(np.uint16)


# ---- moment filters of the fc < 1 correction (tools/sinc3_model.py, NOTES r04): m_i = sum_n (-1)^n win_n (n / 32)^i x[c + n] over ALL
# taps |n| <= 31, i = 0 .. 6.  Same fragment geometry as the bank: pairs (m0 m1), (m2 m3), (m4 m5), (m6 -); fragments 0-2 (m0 m1)
# hi slices 0-2, 3-5 (m0 m1) coefficient lo x 4096, 6-8 (m2 m3), 9-11 (m4 m5), 12-14 (m6 -).  Coefficients below float16's normal
# range are zero to the matrix cores (their whole contribution is < 2e-9 of the peak: sinc3_model.py).
MOMENTS = 7
MOM_FRAGS = ((0, 1, 0, 0), (0, 1, 1, 0), (0, 1, 2, 0), (0, 1, 0, 1), (0, 1, 1, 1), (0, 1, 2, 1),
             (2, 3, 0, 0), (2, 3, 1, 0), (2, 3, 2, 0), (4, 5, 0, 0), (4, 5, 1, 0), (4, 5, 2, 0), (6, -1, 0, 0), (6, -1, 1, 0), (6, -1, 2, 0))


def mom_coef(i, n):
    if i < 0 or abs(n) >= NT:
        return 0.0
    sg = -1.0 if (abs(n) & 1) else 1.0
    return sg * win_n(abs(n)) * (n / 32.0) ** i


def moment_fragments():
    out = np.zeros((len(MOM_FRAGS), 64, 8), dtype=np.float16)
    for fr, (fe, fd, ks, lo) in enumerate(MOM_FRAGS):
        for lane in range(64):
            m, g = lane & 15, lane >> 4
            i, fsel = m >> 1, m & 1
            for j in range(8):
                n = 32 * ks + 8 * g + j - 31 - i
                cf = mom_coef(fd if fsel else fe, n) if -NT < n < NT else 0.0
                hi = np.float16(cf if abs(cf) >= 2.0 ** -14 else 0.0)
                out[fr, lane, j] = np.float16((cf - float(hi)) * LO) if lo else hi
    # the lo fragments must themselves be normal float16 values or zero (the matrix cores flush the rest): those below are dropped
    sub = (np.abs(out.astype(np.float64)) < 2.0 ** -14) & (out != 0)
    out[sub] = 0
    return out.view(np.uint16)


def h16(x):
    """float16 as the matrix cores see it: subnormals flush to zero"""
    v = np.asarray(x, dtype=np.float64).astype(np.float16).astype(np.float64)
    return np.where(np.abs(v) < 2.0 ** -14, 0.0, v)


def split16(x):
    """hi + lo / 4096 with the hi part zeroed below float16's normal range (the lo part then carries the value)"""
    x = np.asarray(x, dtype=np.float64)
    hi = h16(np.where(np.abs(x) < 2.0 ** -14, 0.0, x))
    lo = h16((x - hi) * LO)
    return hi, lo


def filt(f, y_hi, y_lo, c, prods):
    """sum_n coef(f, n) y[c + n] with float16 operands: prods = 3 (hi hi + lo hi + hi lo), 2 (hi hi + hi lo), 1 (hi hi)."""
    ns = np.arange(-31, 32)
    cf = np.array([coef(f, int(n)) for n in ns])
    chi, clo = split16(cf)
    acc = np.sum(chi * y_hi[c + ns])
    if prods >= 2:
        acc += np.sum(chi * y_lo[c + ns]) / LO
    if prods >= 3:
        acc += np.sum(clo * y_hi[c + ns]) / LO
    return acc


def sincw(x, c, s, fc, ks):
    w = WIN[ks + NT] * np.sinc((ks - s) * fc) * fc
    return np.sum(w * x[c + ks])


def model_output(x, img, c, s, fc, K, g0):
    """what the kernel computes for one output.  img: dict of float16-split images (unity: 'x'; general: 'A', 'B')."""
    q = s * s
    ks_near = np.arange(-NEAR, NEAR + 1)
    near = sincw(x, c, s, fc, ks_near)

    def U(key):
        hi, lo = img[key]
        e = [filt(f, hi, lo, c, p) for f, p in ((0, 3), (1, 2), (2, 1))]
        d = [filt(f, hi, lo, c, p) for f, p in ((3, 3), (4, 2), (5, 1))]
        ev = e[0] + q * (e[1] + q * float(np.float16(e[2])))
        dv = d[0] + q * (d[1] + q * float(np.float16(d[2])))
        return (s * ev + dv) / SCALE
    if g0 is None:
        return near - math.sin(math.pi * s) * U('x')
    psi = math.pi * (s - g0 * (K + s))
    eps = fc - (1.0 - g0)
    cp, sp = math.cos(psi), math.sin(psi)
    HA = float(np.float16(filt(6, *img['A'], c, 1)))
    HB = float(np.float16(filt(6, *img['B'], c, 1)))
    H1A = float(np.float16(filt(7, *img['A'], c, 1))) - s * HA
    H1B = float(np.float16(filt(7, *img['B'], c, 1))) - s * HB
    far = -cp * U('A') - sp * U('B') + eps * (cp * HB - sp * HA) + (math.pi * eps * eps / 2) * (cp * H1A + sp * H1B)
    return near + far


def run_case(name, x, speed0, dspeed, n_out=124, g0_off=0.0, seed_pos=0.37):
    """one pass: n_out outputs, speed ramps linearly; returns max |model - reference| / peak."""
    sp = speed0 + dspeed * np.arange(n_out + 1)
    per = 1.0 / sp
    p = 200.0 + seed_pos + np.concatenate(([0.0], np.cumsum(per[:-1])))[:n_out]
    fc = np.minimum(1.0 / per[:n_out], 1.0)
    c = np.rint(p).astype(int)
    s = p - c
    unity = bool(np.all(fc == 1.0))
    ks = np.arange(-NT, NT)
    ref = np.array([sincw(x, c[i], s[i], fc[i], ks) for i in range(n_out)])
    wbase = c[0] - 39
    if unity:
        img = {'x': split16(x)}
        g0 = None
    else:
        fc0 = 0.5 * (fc.min() + fc.max()) + g0_off
        g0 = float(np.float32(1.0 - fc0))
        k = np.arange(len(x)) - wbase
        img = {'A': split16(x * np.sin(np.pi * g0 * k)), 'B': split16(x * np.cos(np.pi * g0 * k))}
    got = np.array([model_output(x, img, c[i], s[i], fc[i], c[i] - wbase, g0) for i in range(n_out)])
    pk = np.max(np.abs(x))
    return float(np.max(np.abs(got - ref)) / pk), unity


def main():
    print("bank: taps %d..31; degrees:" % (NEAR + 1), {d: [n for n in DEGS if DEGS[n] == d] for d in (0, 1, 2)})
    print("worst-case polynomial budgets used (constant, linear, quadratic fits): %.2e %.2e %.2e of the peak" % tuple(BUDGET))
    assert sum(BUDGET) < 1.5e-6
    fr = bank_fragments()
    print("fragments:", fr.shape, "max |coef| e/d:", max(abs(coef(f, n)) for f in range(6) for n in range(-31, 32)))
    rng = np.random.default_rng(3)
    t = np.arange(1200)
    sigs = {"noise": rng.standard_normal(1200), "nyquist": np.cos(np.pi * t), "0.45fs": np.cos(0.9 * np.pi * t + 0.2),
            "fs/4": np.cos(0.5 * np.pi * t + 0.1), "quiet+loud": np.where(t < 600, 1e-3, 1.0) * rng.standard_normal(1200),
            "tiny values": 3e-5 * rng.standard_normal(1200) + (t % 50 == 0)}
    ds = 0.01 * 2 * np.pi * 0.55 / 192000.0
    worst = 0.0
    for name, x in sigs.items():
        for sp0, d, off in ((1.005, ds, 0.0), (1.0, 0.0, 0.0), (1.03, 3 * ds, 0.0), (0.995, ds, 0.0), (0.99, 1e-9, 1.0e-4),
                            (0.97, 3 * ds, -1.5e-4), (0.9995, ds, 1.5e-4), (0.9999, 3 * ds, 0.0)):
            e, unity = run_case(name, x, sp0, d, g0_off=off)
            worst = max(worst, e)
            print(f"{name:10s} speed {sp0:<7} ramp {d:.1e} g0 off {off:+.1e} {'unity  ' if unity else 'general'} max err/peak {e:.2e}")
    print("worst", worst)
    assert worst < 3.0e-6


if __name__ == "__main__":
    main()
