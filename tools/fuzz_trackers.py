#!/usr/bin/env python3
"""Randomised fuzz of the trackers against the numpy oracle (GPU box): random spectrogram sizes, trails (2-5 points,
unsorted, starting at 0 or inside the file), tolerances, FM pilots with noise; Peak / Peak Track / Center of Gravity
on the device (K_track), Correlation (device, par_track_corr_f64) and Zero-Crossing (device filter + compaction)."""
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import oracle_np as O
from pyaudiorestoration_amd import fourier as F, wow_detection as W

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t_end = time.time() + budget
case = 0
worst = {}
warnings.simplefilter("ignore")
while time.time() < t_end:
    rng = np.random.default_rng(case)
    sr = int(rng.choice([44100, 96000, 192000]))
    n_fft = int(rng.choice([256, 512, 1024, 2048]))
    hop = int(rng.choice([n_fft // 8, n_fft // 4, n_fft // 2]))
    dur = float(rng.uniform(0.4, 2.0))
    n = int(sr * dur)
    f0 = float(rng.uniform(800, min(12000, sr / 4)))
    t = np.arange(n) / sr
    phase = 2 * np.pi * f0 * t + (f0 * rng.uniform(0.001, 0.01) / rng.uniform(2, 12)) * np.sin(2 * np.pi * rng.uniform(2, 12) * t)
    x = (np.sin(phase) + 10 ** rng.uniform(-4, -1) * rng.standard_normal(n)).astype(np.float32)
    mag_t = F.get_mag(torch.from_numpy(x).cuda(), n_fft, hop, "blackmanharris", 1)
    spec = mag_t.cpu().numpy()        # the SAME spectrogram for both sides: band edges sit on rounding cliffs, and a
    #                                   1e-7 difference between two STFTs would be amplified into a different band
    npts = int(rng.integers(2, 6))
    ts = rng.uniform(0.02 * dur, 0.98 * dur, npts)
    if rng.random() < 0.2:
        ts[0] = 0.0
    trail = [(float(a), float(f0 * rng.uniform(0.995, 1.005))) for a in ts]
    tol = float(rng.choice([0.2, 0.5, 1.0, 3.0]))
    for name in ("Peak", "Peak Track", "Center of Gravity", "Correlation", "Freehand Draw"):
        try:
            t_ref, f_ref = O.TRACKERS[name](spec, list(trail), n_fft, hop, sr, tol)
        except Exception as e:
            try:
                W.wow_detectors[name](mag_t, x[:, None], list(trail), n_fft, hop, sr, tol, "Linear")
            except Exception as e2:
                assert isinstance(e2, type(e)) or name == "Correlation", (case, name, repr(e), repr(e2))
                continue
            raise SystemExit(f"case {case} {name}: oracle raised {e!r}, device path did not")
        tr = W.wow_detectors[name](mag_t, x[:, None], list(trail), n_fft, hop, sr, tol, "Linear")
        assert np.array_equal(tr.times, t_ref), (case, name, "times")
        err = float(np.max(np.abs(tr.freqs - f_ref)) / np.max(np.abs(f_ref))) if len(f_ref) else 0.0
        lim = 1e-5 if name != "Correlation" else 1e-4
        assert err < lim, (case, name, err, sr, n_fft, hop, tol, trail)
        worst[name] = max(worst.get(name, 0.0), err)
    try:
        t_ref, f_ref = O.track_zero_crossing(spec, x[:, None], list(trail), n_fft, hop, sr, tol)
    except Exception:
        t_ref = None
    if t_ref is not None and len(t_ref) > 4:
        tr = W.wow_detectors["Zero-Crossing"](mag_t, x[:, None], list(trail), n_fft, hop, sr, tol, "Linear")
        err = float(np.max(np.abs(tr.freqs - f_ref)) / np.max(np.abs(f_ref)))
        assert np.array_equal(tr.times, t_ref) and err < 1e-6, (case, "Zero-Crossing", err)
        worst["Zero-Crossing"] = max(worst.get("Zero-Crossing", 0.0), err)
    case += 1
print(f"tracker fuzz ok: {case} cases, worst relative errors {worst}")
