#!/usr/bin/env python3
"""Fuzz of the config-4 chain (GPU box): random signals, fft sizes / hops, 1-12 random dropout boxes (overlapping,
nested, tiny, wide) against the oracle's serial marker loop; boxes that leave the spectrogram must be refused with
ValueError, never crash."""
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import oracle_np as O
from pyaudiorestoration_amd import pipeline as P

warnings.simplefilter("ignore")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t_end = time.time() + budget
case = refused = sparse_cases = 0
worst = worst_sd = 0.0
while time.time() < t_end:
    rng = np.random.default_rng(case)
    sr = int(rng.choice([22050, 44100, 96000]))
    n_fft = int(rng.choice([256, 512, 1024]))
    hop = int(rng.choice([n_fft // 16, n_fft // 8, n_fft // 4]))
    n = int(rng.integers(12000, 50000)) if rng.random() < 0.5 else int(rng.integers(150000, 400000))   # long ones: the sparse path
    x = (0.3 * np.sin(2 * np.pi * rng.uniform(300, 5000) * np.arange(n) / sr) + 0.05 * rng.standard_normal(n)).astype(np.float32)
    dur = n / sr
    marks = []
    for _ in range(int(rng.integers(1, 13))):
        t0 = rng.uniform(-0.02, dur + 0.02) if rng.random() < 0.1 else rng.uniform(0.1 * dur, 0.9 * dur)
        w = rng.uniform(0.001, 0.05)
        f0 = rng.uniform(50, sr / 2 * 0.8)
        f1 = f0 + rng.uniform(50, sr / 2 - f0)
        marks.append((t0, f0, t0 + w, f1, float(rng.choice([0.1, 0.5, 1.0]))))
        k = slice(max(0, int(t0 * sr)), max(0, int((t0 + w) * sr)))
        x[k] *= np.float32(0.1)
    geo = [P.marker_geometry(m, sr, hop, n_fft) for m in marks]
    frames = (n + n_fft // 2) // hop + 1
    valid = all(fb - fs >= 0 and fa + fs <= frames and fa - fb >= 1 and bu - bl >= 1 for fb, fa, fs, bl, bu in geo)
    if not valid:
        try:
            P.heal_dropouts(x, sr, marks, n_fft, hop)
        except ValueError:
            refused += 1
            case += 1
            continue
        raise SystemExit(f"case {case}: a box outside the spectrogram was accepted")
    want = O.heal_dropouts(x, sr, marks, n_fft, hop)
    got = P.heal_dropouts(x, sr, marks, n_fft, hop)                      # sparse where the boxes leave most of the file alone (r03)
    dense = P.heal_dropouts(x, sr, marks, n_fft, hop, sparse=False)      # the reference-shaped dataflow
    scale = max(float(np.max(np.abs(want))), 1e-30)
    if P.heal_segments(geo, frames, n + n_fft // 2, n, n_fft, hop) is not None:
        sparse_cases += 1
        worst_sd = max(worst_sd, float(np.max(np.abs(got - dense)) / scale))
    err = max(float(np.max(np.abs(got - want)) / scale), float(np.max(np.abs(dense - want)) / scale))
    # the boost (up to tens of dB) multiplies the ~1e-7 float32 rounding noise of either STFT: 1e-5 is reached at ~40 dB
    assert np.all(np.isfinite(got)) and err < 5e-5, (case, err, sr, n_fft, hop, n, marks)
    worst = max(worst, err)
    case += 1
print(f"heal fuzz ok: {case} cases ({refused} with out-of-range boxes refused, {sparse_cases} through the sparse path), worst relative "
      f"error {worst:.2e}, worst sparse - dense difference {worst_sd:.2e}")
