#!/usr/bin/env python3
"""Fuzz of the operator-slot entry points with ARBITRARY caller positions (GPU box): sinc_wrapper(sample_at, ...) and
the Linear mode accept any float64 array -- non-monotonic, repeated, negative, past the end, huge -- not only what
speed_to_pos produces.  K_sinc against the C oracle, K_lerp against np.interp."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import oracle_c as C, oracle_np as O
from pyaudiorestoration_amd import resampling as R

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t_end = time.time() + budget
case, worst = 0, 0.0
while time.time() < t_end:
    rng = np.random.default_rng(case)
    n = int(rng.choice([40, 500, 5000, 60000]))
    k = int(rng.choice([2, 3, 100, 3000, 40000]))
    NT = int(rng.choice([1, 4, 16, 32, 50]))
    sig = rng.standard_normal(n).astype(np.float32)
    style = int(rng.integers(0, 6))
    if style == 0:
        pos = rng.uniform(-60, n + 60, k)                              # random order, off both ends
    elif style == 1:
        pos = np.sort(rng.uniform(0, n, k))[::-1].copy()               # decreasing
    elif style == 2:
        pos = np.repeat(rng.uniform(0, n, (k + 3) // 4), 4)[:k]       # repeated positions (dp == 0)
    elif style == 3:
        pos = np.cumsum(rng.uniform(0.2, 3.0, k))                      # monotone, wide speed range
        pos = pos % (n + 40) - 20
    elif style == 4:
        pos = rng.integers(-5, n + 5, k).astype(np.float64) + rng.choice([0.0, 0.5, -0.5], k)   # exact ties
    else:
        pos = rng.uniform(0, n, k)
        pos[rng.integers(0, k, max(1, k // 50))] = rng.choice([-1e9, 1e9, 1e300, -1e300])      # wild values
    want = C.sinc(pos, sig, NT)
    got = R.sinc_wrapper(pos, sig, 0, NT)
    # relative to the signal level: with windows that barely touch a very short signal every output is tiny
    scale = max(float(np.max(np.abs(want))), float(np.max(np.abs(sig))), 1e-30)
    err = float(np.max(np.abs(got - want)) / scale)
    assert err < 5e-6, (case, "sinc", err, n, k, NT, style)
    lin_want = O.linear_resample(pos, sig)
    lin_got = R.linear_resample_dev(torch.from_numpy(pos).cuda(), torch.from_numpy(sig).cuda()).cpu().numpy()
    assert np.array_equal(lin_got, lin_want), (case, "linear", n, k, style)
    worst = max(worst, err)
    case += 1
print(f"operator-slot fuzz ok: {case} cases, worst sinc relative error {worst:.2e}, linear mode bit-exact")
