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

# ---- flows
from pyaudiorestoration_amd import pipeline
sr, n = 192000, 192000 * 600
x = inputs.pilot(n, sr)
xt = torch.from_numpy(x).cuda()
print(f"pipeline.respeed (10 min @192k mono, Peak, NT 32, numpy in): "
      f"{best(lambda: pipeline.respeed(x, sr, [(1.0, 3950.0), (598.0, 4050.0)], 1024, 256, 1, 'Peak', 0.5, (0, 20), 32), 2):8.1f} ms")
m = n // 256
st = torch.linspace(0, n, m, dtype=torch.float64, device="cuda")
sp = 1.0 + 0.01 * torch.sin(torch.arange(m, dtype=torch.float64, device="cuda") * 0.01)
print(f"speed_to_pos_dev (position array, {n} samples): {best(lambda: resampling.speed_to_pos_dev(st, sp, n)):8.3f} ms")
xd = torch.from_numpy(x.astype(np.float64)).cuda()
print(f"zero_crossings_dev ({n} samples): {best(lambda: wow_detection.zero_crossings_dev(xd)):8.3f} ms")
n4, sr4 = 322531, 44100
x4 = inputs.noise(n4, 3)
rng = np.random.default_rng(4)
marks = []
for t0 in np.sort(rng.uniform(0.2, n4 / sr4 - 0.2, 32)):
    w = rng.uniform(0.004, 0.02)
    marks.append((t0 - w / 2, 500.0, t0 + w / 2, 9000.0, 0.5))
print(f"pipeline.heal_dropouts (config 4 x1, numpy in/out): {best(lambda: pipeline.heal_dropouts(x4[:, None], sr4, marks, 512, 32)):8.3f} ms")
mag4 = fourier.get_mag(torch.from_numpy(x4).cuda(), 512, 32, 'hann', 1)
print(f"pipeline.detect_dropouts (x1): {best(lambda: pipeline.detect_dropouts(mag4, sr4, 512, 32, 0.5, 6.5, 2000.0, 8000.0)):8.3f} ms")
