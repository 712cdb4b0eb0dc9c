"""The MOMENT form of the fc < 1 taps (csrc/sinc2.hip k_sinc_pipe<false, true>, tools/sinc3_model.py) against the reference's own
weights (util/resampling.py:66-87: win_n sinc((n - s) fc) fc over offsets -NT .. NT-1), in numpy on the CPU:
    out(fc) = out(1) - g (cos(pi s) Re Q - sin(pi s) Im Q),   g = 1 - fc,
    Q = sum_i M_i (i 32 pi g)^i (alpha_i + i beta_i),  M_i = sum_n (-1)^n win_n (n / 32)^i x[c + n]   (seven FIXED filters)
The kernel evaluates exactly this (float32, the moments from float16 operands on the matrix cores); the GPU parity of the kernel
is tests/test_hip_parity.py::test_streaming_kernel_opt_in[4].  Here: the identity itself, its validity range, and that the
constant fragments the kernel loads are the ones this model generates."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

NT = 32
N = np.arange(-NT, NT)
WIN = np.hanning(2 * NT + 1)[:2 * NT].astype(np.float32).astype(np.float64)
SGN = np.where(N % 2 == 0, 1.0, -1.0)


def ref_weights(s, fc):
    return WIN * np.sinc(fc * (N - s)) * fc


def moment_correction(m, s, g, f32):
    c = (lambda v: np.float32(v)) if f32 else (lambda v: v)
    G = c(math.pi) * c(g)
    w = G * c(s)
    w2, G32 = w * w, c(32.0) * G
    re, im = c(m[6]) * c(1.0 / (720 * 7)), c(0.0)
    for i in (5, 4, 3, 2, 1, 0):
        fi = math.factorial(i)
        al = c(1.0 / (fi * (i + 1))) - (w2 * c(1.0 / (2 * fi * (i + 3))) if i <= 2 else c(0.0))
        be = -w * c(1.0 / (fi * (i + 2)))
        re, im = c(m[i]) * al - G32 * im, c(m[i]) * be + G32 * re
    return -c(g) * (c(math.cos(math.pi * s)) * re - c(math.sin(math.pi * s)) * im)


def worst(gmax, trials, f32, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(4096)
    out = 0.0
    for x in (rng.standard_normal(4096), np.cos(np.pi * t), np.cos(0.98 * np.pi * t + 1.0), np.cos(0.9 * np.pi * t + 0.2)):
        x = x.astype(np.float32).astype(np.float64)
        for _ in range(trials):
            c0 = int(rng.integers(NT, len(x) - NT))
            s, g = float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0.0, gmax))
            xs = x[c0 + N]
            want = float(np.dot(ref_weights(s, 1.0 - g), xs))
            m = [float(np.dot(SGN * WIN * (N / 32.0) ** i, xs)) for i in range(7)]
            got = float(np.dot(ref_weights(s, 1.0), xs)) + float(moment_correction(m, s, g, f32))
            out = max(out, abs(got - want) / np.max(np.abs(x)))
    return out


def test_moment_identity_holds_to_1e7_on_the_benchmark_range():
    assert worst(0.0101, 300, False, 1) < 2.5e-7
    assert worst(0.0101, 300, True, 2) < 4e-7            # float32 evaluation, as in the kernel


def test_moment_identity_validity_limit_is_where_the_kernel_puts_it():
    # the kernel sends tiles with 1 - fc > 0.0125 to the block kernel (kEpMaxMom): inside, the seven moments hold 1e-6
    assert worst(0.0125, 300, True, 3) < 1.5e-6
    assert worst(0.03, 200, False, 4) > 1e-5              # far outside, they do not: the limit is needed


def test_moment_fragments_in_the_header_are_the_models():
    import re
    import sinc2_model as M2
    fr = M2.moment_fragments().reshape(-1, 2)
    words = (fr[:, 0].astype(np.uint32) | (fr[:, 1].astype(np.uint32) << 16))
    text = open(os.path.join(ROOT, "pyaudiorestoration_amd", "csrc", "sinc_taps_gen.h")).read()
    body = text[text.index("kBank3Frags32["):]
    body = body[body.index("{") + 1:body.index("};")]
    have = np.array([int(h, 16) for h in re.findall(r"0x([0-9a-f]{8})u", body)], dtype=np.uint32)
    assert have.shape == words.shape == (15 * 64 * 4,) and np.array_equal(have, words)
    # coefficient (0, n) of the first fragment pair is the alternating Hann window itself
    assert abs(M2.mom_coef(0, 3) + float(WIN[NT + 3])) < 1e-12 and M2.mom_coef(1, 0) == 0.0
