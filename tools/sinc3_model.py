"""Numerical model of the MOMENT form of the fc < 1 taps (NOTES r04; not built as a kernel yet).
  w_n(fc) = win_n sin(pi fc (n - s)) / (pi (n - s)),  g = 1 - fc,  phi = pi g (n - s):
  w_n(fc) - w_n(1) = g (-1)^n win_n [ -sin(pi s) a(phi) - cos(pi s) b(phi) ],   a = (cos phi - 1) / phi,  b = sin(phi) / phi
  a, b are entire: with G = pi g and the shifted moments  P_k(s) = sum_n (-1)^n win_n (n - s)^k x[c+n]
      out(fc) = out(1) - g [ sin(pi s) sum_{k odd} a_k G^k P_k + cos(pi s) sum_{k even} b_k G^k P_k ]
  and P_k(s) = sum_i C(k,i) (-s)^(k-i) m_i with the FIXED moment filters m_i = sum_n (-1)^n win_n n^i x[c+n] (all 63 taps).
This script checks (1) the truncation order K against exact weights over g <= gmax, (2) the float16 budget: signal as hi + lo
2^-12, moment coefficients (n/32)^i win_n as float16 (hi only / hi + lo), which products have to be kept."""
import math, sys
import numpy as np
NT = 32
n = np.arange(-NT, NT)                                   # the reference's window: offsets -NT .. NT-1
WIN = np.hanning(2 * NT + 1)[:2 * NT].astype(np.float32).astype(np.float64)
SGN = np.where(n % 2 == 0, 1.0, -1.0)

def exact_weights(s, fc):
    u = n - s
    return WIN * np.sinc(fc * u) * fc                    # sin(pi fc u) / (pi u)

def series(K):
    # a(phi) = sum_{j>=1} (-1)^j phi^(2j-1) / (2j)!,   b(phi) = sum_{j>=0} (-1)^j phi^(2j) / (2j+1)!
    a = {2 * j - 1: (-1) ** j / math.factorial(2 * j) for j in range(1, K + 2) if 2 * j - 1 <= K}
    b = {2 * j: (-1) ** j / math.factorial(2 * j + 1) for j in range(0, K + 2) if 2 * j <= K}
    return a, b

def f16(x):
    return np.asarray(x, dtype=np.float64).astype(np.float16).astype(np.float64)
def f16_flush(x):                                        # what the matrix cores see: subnormal float16 operands are zero
    h = f16(x)
    return np.where(np.abs(h) < 6.103515625e-05, 0.0, h)

def run(K, gmax, trials, rng, mode):
    a, b = series(K)
    worst = {}
    for name, gen in (("noise", lambda m: rng.standard_normal(m)), ("nyquist", lambda m: np.cos(np.pi * np.arange(m))),
                      ("0.45 fs", lambda m: np.cos(0.9 * np.pi * np.arange(m) + 0.2)), ("0.49 fs", lambda m: np.cos(0.98 * np.pi * np.arange(m) + 1.0))):
        x = gen(4096).astype(np.float32).astype(np.float64)
        pk = np.max(np.abs(x))
        w = 0.0
        for _ in range(trials):
            c = int(rng.integers(NT, len(x) - NT))
            s = float(rng.uniform(-0.5, 0.5))
            g = float(rng.uniform(0.0, gmax))
            xs = x[c + n]
            ref = float(np.dot(exact_weights(s, 1.0 - g), xs))
            unity = float(np.dot(exact_weights(s, 1.0), xs))
            # fixed moment filters, coefficients scaled by (n / 32)^i
            if mode == "f64":
                m = [float(np.dot(SGN * WIN * (n / 32.0) ** i, xs)) for i in range(K + 1)]
            else:
                hi = f16_flush(xs)
                lo = f16((xs - hi) * 4096.0)
                m = []
                for i in range(K + 1):
                    co = SGN * WIN * (n / 32.0) ** i
                    ch = f16_flush(co)
                    cl = f16_flush((co - ch) * 4096.0)
                    acc = np.dot(ch, hi)                                      # hi x hi (float32 accumulation is ample: modelled exact)
                    if i <= LO_SIG: acc += np.dot(ch, lo) / 4096.0            # signal lo
                    if i <= LO_COEF: acc += np.dot(cl, hi) / 4096.0           # coefficient lo
                    m.append(float(acc))
            G = math.pi * g
            S, C = math.sin(math.pi * s), math.cos(math.pi * s)
            def P(k):                                                         # shifted moment of order k from the fixed ones
                return sum(math.comb(k, i) * (-s) ** (k - i) * m[i] * 32.0 ** i for i in range(k + 1))
            corr = -g * (S * sum(ak * G ** k * P(k) for k, ak in a.items()) + C * sum(bk * G ** k * P(k) for k, bk in b.items()))
            w = max(w, abs(unity + corr - ref) / pk)
        worst[name] = w
    return worst

rng = np.random.default_rng(3)
print("truncation (float64 arithmetic): worst |error| / peak over 4000 random (centre, s, g <= gmax)")
for gmax in (0.0101, 0.02, 0.03):
    for K in (4, 5, 6, 7, 8, 10):
        r = run(K, gmax, 4000, rng, "f64")
        print(f"  gmax {gmax:.4f}  K {K:2d}  " + "  ".join(f"{k} {v:.1e}" for k, v in r.items()))
print("float16 operands, K = 6, gmax 0.0101: which lo parts are kept (signal lo for i <= LO_SIG, coefficient lo for i <= LO_COEF)")
for LO_SIG, LO_COEF in ((-1, -1), (0, -1), (1, -1), (1, 0), (1, 1), (3, 1), (3, 3), (6, 6)):
    r = run(6, 0.0101, 4000, rng, "f16")
    print(f"  signal lo <= {LO_SIG:2d}, coefficient lo <= {LO_COEF:2d}:  " + "  ".join(f"{k} {v:.1e}" for k, v in r.items()))

# ---- the form a kernel would evaluate per output: no triangular shift of the moments.  With z = i G, u = n - s:
#   sum_k z^k P_k / (k+1)! = sum_n wn x_n (e^{z u} - 1) / (z u) = int_0^1 e^{-t z s} M(t z) dt,   M(w) = sum_i m_i w^i / i!
#   |z s| <= 0.016: e^{-t z s} to second order, int_0^1 t^i (1 - t z s + t^2 (z s)^2 / 2) dt = 1/(i+1) - z s/(i+2) + (z s)^2 / (2 (i+3))
#   Q = sum_i m_i (i G)^i (alpha_i + i beta_i),  alpha_i = 1/(i! (i+1)) - w^2 / (2 i! (i+3)),  beta_i = -w / (i! (i+2)),  w = G s
#   out(fc) = out(1) - g (cos(pi s) Re Q - sin(pi s) Im Q)
def kernel_form(m, s, g, K, f32=False):
    cast = (lambda v: np.float32(v)) if f32 else (lambda v: v)
    G = cast(math.pi) * cast(g)
    w = G * cast(s)
    w2 = w * w
    G32 = cast(32.0) * G
    re = cast(0.0); im = cast(0.0)
    # Horner in G32 from the top moment down: Q = sum_i M_i (i G32)^i (alpha_i + i beta_i)
    for i in range(K, -1, -1):
        fi = math.factorial(i)
        al = cast(1.0 / (fi * (i + 1))) - w2 * cast(1.0 / (2 * fi * (i + 3)))
        be = -w * cast(1.0 / (fi * (i + 2)))
        tr, ti = cast(m[i]) * al, cast(m[i]) * be
        # Q_i = t + (i G32) Q_{i+1}:  (re, im) <- (tr - G32 im, ti + G32 re)
        re, im = tr - G32 * im, ti + G32 * re
    S, C = cast(math.sin(math.pi * s)), cast(math.cos(math.pi * s))
    return -cast(g) * (C * re - S * im)

def check_kernel_form(K=6, gmax=0.0101, trials=6000):
    rng = np.random.default_rng(11)
    for name, gen in (("noise", lambda k: rng.standard_normal(k)), ("nyquist", lambda k: np.cos(np.pi * np.arange(k))),
                      ("0.49 fs", lambda k: np.cos(0.98 * np.pi * np.arange(k) + 1.0)), ("0.45 fs", lambda k: np.cos(0.9 * np.pi * np.arange(k) + 0.2))):
        x = gen(4096).astype(np.float32).astype(np.float64)
        pk = np.max(np.abs(x))
        w64 = w32 = 0.0
        for _ in range(trials):
            c = int(rng.integers(NT, len(x) - NT)); s = float(rng.uniform(-0.5, 0.5)); g = float(rng.uniform(0.0, gmax))
            xs = x[c + n]
            ref = float(np.dot(exact_weights(s, 1.0 - g), xs)) - float(np.dot(exact_weights(s, 1.0), xs))
            m = [float(np.dot(SGN * WIN * (n / 32.0) ** i, xs)) for i in range(K + 1)]
            w64 = max(w64, abs(kernel_form(m, s, g, K) - ref) / pk)
            w32 = max(w32, abs(float(kernel_form(m, s, g, K, f32=True)) - ref) / pk)
        print(f"  kernel form K {K} gmax {gmax}: {name:8s} float64 {w64:.1e}  float32 {w32:.1e}")

print("per-output closed form (second order in G s), exact moments:")
check_kernel_form(6, 0.0101)
check_kernel_form(6, 0.0125)
check_kernel_form(5, 0.0101)
