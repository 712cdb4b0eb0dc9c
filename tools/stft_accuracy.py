#!/usr/bin/env python3
"""K_stft accuracy against a float64 evaluation of the same framing, and what it does to the config-3 tracker:
prints max/rms relative error of the magnitude spectrogram (relative to its peak) and the PeakTracker frequency
deviation from the float64 evaluation, for the pilot workload of tests/golden/pipeline.npz."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scipy.signal

import inputs
from oracle import oracle_np as O
from pyaudiorestoration_amd import fourier

g = np.load(os.path.join(ROOT, "tests", "golden", "pipeline.npz"))
sr, n, n_fft, hop = (int(v) for v in g["cfg"])
x = inputs.pilot(n, sr)
trail = [(0.05, 4000.0), (1.45, 4000.0)]
win = scipy.signal.get_window("blackmanharris", n_fft).astype(np.float32).astype(np.float64)
xp = np.pad(x.astype(np.float64), n_fft // 2, mode="reflect")
nf = (len(xp) - n_fft) // hop + 1
fr = np.stack([xp[i * hop:i * hop + n_fft] * win for i in range(nf)], axis=1)
mag64 = np.abs(np.fft.rfft(fr, axis=0) / np.sqrt(n_fft)) + 1e-7
mag_ref = O.get_mag(x, n_fft, hop, "blackmanharris")
mag_gpu = np.asarray(fourier.get_mag(x, n_fft, hop, "blackmanharris"))
pk = mag64.max()
for name, m in (("reference float32 numpy path", mag_ref), ("K_stft", mag_gpu)):
    e = (m - mag64) / pk
    _, f = O.track_peak(m, trail, n_fft, hop, sr, 0.5)
    _, f64 = O.track_peak(mag64, trail, n_fft, hop, sr, 0.5)
    print(f"{name:30s} magnitude error vs float64: max {np.max(np.abs(e)):.2e} rms {np.sqrt(np.mean(e * e)):.2e}   "
          f"tracked-frequency deviation: max {np.max(np.abs(f - f64) / f64):.2e}")
