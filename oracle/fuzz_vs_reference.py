#!/usr/bin/env python3
"""Pins the ORACLE against the real reference on random inputs (runs only where /root/reference is mounted; nothing
here travels to the GPU box).  The golden fixtures pin the oracle on fixed cases; this widens the pin: random speed
curves, grids, signal lengths, NT, STFT sizes, filter designs and tracker trails go through the reference's own
functions (imported with the numba/soundfile stubs of gen_golden.py) and through oracle_np / oracle_c.

    python oracle/fuzz_vs_reference.py [seconds] [--ref /root/reference]
"""
import argparse
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from gen_golden import import_reference, written_len  # noqa: E402
from oracle import oracle_c as C, oracle_np as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("seconds", nargs="?", type=float, default=60.0)
ap.add_argument("--ref", default="/root/reference")
a = ap.parse_args()
fourier, resampling, filters, wow, correlation = import_reference(a.ref)
warnings.simplefilter("ignore")
t_end = time.time() + a.seconds
case = 0
stats = {"pos": 0, "pos_refused": 0, "sinc": 0, "stft": 0, "istft": 0, "filt": 0, "track": 0}


def relerr(x, y):
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    return float(np.max(np.abs(x - y)) / max(float(np.max(np.abs(y))), 1e-300)) if x.size else 0.0


while time.time() < t_end:
    rng = np.random.default_rng(case)
    # ---- speed_to_pos: value- and refusal-parity, numpy and C restatements ---------------------------------
    n = int(rng.choice([600, 3000, 20000]))
    seg = int(rng.choice([8, 16, 64, 256, 1000]))
    m = max(2, n // seg)
    st = np.linspace(0, n, m) + float(rng.choice([0.0, 0.0, 123.456, -50.0]))
    style = int(rng.integers(0, 5))
    sp = [1.0 + 0.01 * np.sin(np.arange(m) * 0.3 + case), rng.uniform(0.5, 2.0, m), np.ones(m),
          np.where(rng.random(m) < 0.5, 2.0 / seg, 1.0) * rng.uniform(0.99, 1.01, m),
          rng.choice([0.05, 8.0]) * rng.uniform(0.9, 1.1, m)][style]
    try:
        ref = resampling.speed_to_pos(st, sp, n)
        ref = ref[:written_len(st, sp, len(ref))]
    except Exception as e_ref:
        for name, fn in (("numpy oracle", lambda: O.speed_to_pos(st, sp, n)), ("C oracle", lambda: C.speed_to_pos(st, sp, n))):
            try:
                fn()
            except Exception:
                continue
            raise SystemExit(f"case {case}: the reference raised {e_ref!r} but the {name} accepted the curve")
        stats["pos_refused"] += 1
        ref = None
    if ref is not None and np.isnan(ref).any():
        # a 1-sample segment: the reference divides 0/0 and returns NaN positions from there on.  The build's decision
        # (DESIGN section 2 / INTEGRATION): the C oracle and the device refuse such a curve with that diagnosis.
        try:
            C.speed_to_pos(st, sp, n)
        except ValueError:
            stats["pos_refused"] += 1
            ref = None
        else:
            raise SystemExit(f"case {case}: NaN positions in the reference but the C oracle accepted the curve")
    if ref is not None:
        got_np, _ = O.speed_to_pos(st, sp, n)
        got_c, _ = C.speed_to_pos(st, sp, n)
        assert np.array_equal(got_np, ref) and np.array_equal(got_c, ref), (case, "speed_to_pos", n, seg, style)
        stats["pos"] += 1
        # ---- sinc_core on a short stretch of those positions (the reference runs ~1e5 samples/s without numba)
        if len(ref) > 40:
            NT = int(rng.choice([1, 4, 16, 32, 50]))
            sig = rng.standard_normal(n).astype(np.float32)
            lo = int(rng.integers(0, len(ref) - 30))
            pos = ref[lo:lo + int(rng.integers(2, 400))]
            try:
                want = resampling.sinc_wrapper(pos, sig, 0, NT)
            except ValueError:
                # position below -NT: the reference's slice end min(ind+NT, len) goes negative, Python reads it as
                # "from the end", and the tap product fails to broadcast.  resampling.run never produces such
                # positions (it clips at 0, :205); the build returns 0.0 there (empty window).
                assert pos.min() < 0, (case, "sinc raised on non-negative positions")
                want = None
            assert want is None or relerr(O.sinc_resample(pos, sig, NT), want) < 2e-7 and relerr(C.sinc(pos, sig, NT), want) < 2e-7, (case, "sinc", NT)
            stats["sinc"] += 1
    # ---- stft / get_mag / istft --------------------------------------------------------------------------
    n_fft = int(2 ** rng.integers(4, 12))
    hop = max(1, int(rng.choice([n_fft // 8, n_fft // 4, n_fft // 2, n_fft])))
    x = rng.standard_normal(int(rng.choice([n_fft + 3, 5 * n_fft + 1, 12000]))).astype(np.float32)
    win = str(rng.choice(["hann", "blackmanharris", "hamming"]))
    zp = int(rng.choice([1, 1, 2]))
    S = fourier.stft(x, n_fft, hop, win, zp)
    assert relerr(O.stft(x, n_fft, hop, win, zp), S) < 1e-6 and relerr(O.get_mag(x, n_fft, hop, win, zp), fourier.get_mag(x, n_fft, hop, win, zp)) < 1e-6
    stats["stft"] += 1
    if zp == 1 and 2 * hop <= n_fft:
        Sc = np.array(S).astype(np.complex64)
        want = fourier.istft(Sc.copy(), hop_length=hop, window_name=win, length=len(x))
        assert relerr(O.istft(Sc, hop, win, len(x)), want) < 1e-5, (case, "istft", n_fft, hop)
        stats["istft"] += 1
    # ---- filters -----------------------------------------------------------------------------------------
    fs = float(rng.choice([172.265625, 44100.0]))
    lo_c, hi_c = float(rng.choice([0.0, 0.01, 0.2])) * fs / 2, float(rng.choice([0.0, 0.3, 0.9, 1.5])) * fs / 2
    y = rng.standard_normal(int(rng.choice([40, 500, 5000])))
    order = int(rng.integers(1, 6))
    try:
        want = filters.butter_bandpass_filter(y, lo_c, hi_c, fs, order=order)
    except ValueError as e:
        try:
            O.butter_bandpass_filter(y, lo_c, hi_c, fs, order=order)
        except ValueError as e2:
            assert str(e) == str(e2)
        else:
            raise SystemExit(f"case {case}: filter refusal not reproduced")
    else:
        got = O.butter_bandpass_filter(y, lo_c, hi_c, fs, order=order)
        assert (got is y) == (want is y) and relerr(got, want) < 1e-12, (case, "filter")
    stats["filt"] += 1
    # ---- trackers ----------------------------------------------------------------------------------------
    sr = 48000
    tt = np.arange(30000) / sr
    pilot = np.sin(2 * np.pi * 3000 * tt + 2.0 * np.sin(2 * np.pi * 6 * tt)).astype(np.float32)
    spec = fourier.get_mag(pilot, 512, 128, "blackmanharris", 1)
    trail = [(float(rng.uniform(0.02, 0.2)), 3000.0), (float(rng.uniform(0.3, 0.6)), float(rng.uniform(2990, 3010)))]
    tol = float(rng.choice([0.3, 0.5, 1.0]))
    for name in ("Peak", "Peak Track", "Center of Gravity", "Correlation", "Freehand Draw"):
        tr = wow.wow_detectors[name](spec, pilot[:, None], list(trail), 512, 128, sr, tol, "Linear")
        t_o, f_o = O.TRACKERS[name](spec, list(trail), 512, 128, sr, tol)
        assert np.array_equal(t_o, tr.times) and relerr(f_o, tr.freqs) < 1e-9, (case, name)
    tr = wow.wow_detectors["Zero-Crossing"](spec, pilot[:, None], list(trail), 512, 128, sr, tol, "Linear")
    t_o, f_o = O.track_zero_crossing(spec, pilot[:, None], list(trail), 512, 128, sr, tol)
    assert np.array_equal(t_o, tr.times) and relerr(f_o, tr.freqs) < 1e-9, (case, "Zero-Crossing")
    stats["track"] += 1
    case += 1
print(f"oracle == reference on {case} random cases: {stats}")
