#!/usr/bin/env python3
"""Timing sanity of the secondary device paths (looking for cliffs, not records): correlation, trackers, filters, ISTFT
sizes, linear / lag resampling.  Best of 3, inputs resident where the API allows."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import inputs
from pyaudiorestoration_amd import _dev, correlation, filters, fourier, resampling, wow_detection


def best(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        b = min(b, time.perf_counter() - t0)
    return b * 1e3


rng = np.random.default_rng(0)
for n in (1 << 14, 1 << 17, 1 << 19):
    a = torch.from_numpy(rng.standard_normal(n)).cuda()
    b = torch.roll(a, 37)
    print(f"xcorr_dev 2 x {n:8d}: {best(lambda: correlation.xcorr_dev(a, b)):8.3f} ms")
    an, bn = a.cpu().numpy().copy(), b.cpu().numpy().copy()
    print(f"find_delay 2 x {n:8d} (host arrays in): {best(lambda: correlation.find_delay(an, bn)):8.3f} ms")
for n in (10 ** 5, 10 ** 7, 10 ** 8):
    x = torch.from_numpy(rng.standard_normal(n)).cuda()
    print(f"bandpass_dev order 3, {n:10d} samples: {best(lambda: filters.bandpass_dev(x, 300.0, 6000.0, 48000.0, 3)):8.3f} ms")
sr, n = 192000, 192000 * 120
x = inputs.pilot(n, sr)
xt = torch.from_numpy(x).cuda()
mag = fourier.get_mag(xt, 4096, 1024, "blackmanharris", 1)
trail = [(1.0, 3950.0), (118.0, 4050.0)]
for name in ("Peak", "Peak Track", "Center of Gravity", "Correlation", "Zero-Crossing", "Partials"):
    print(f"tracker {name:18s} on {mag.shape[1]} frames x {mag.shape[0]} bins: "
          f"{best(lambda: wow_detection.wow_detectors[name](mag, x[:, None], list(trail), 4096, 1024, sr, 0.5, 'Linear')):8.3f} ms")
for n_fft, hop in ((512, 32), (1024, 256), (4096, 1024), (8192, 2048), (8192, 512)):
    S = fourier.stft(xt, n_fft, hop)
    print(f"istft {n_fft}/{hop} of {S.shape[1]} frames: {best(lambda: fourier.istft(S, hop_length=hop, length=n)):8.3f} ms")
pos = torch.arange(n, dtype=torch.float64, device="cuda") * 0.999
print(f"linear_resample_dev {n} outputs: {best(lambda: resampling.linear_resample_dev(pos, xt)):8.3f} ms")
