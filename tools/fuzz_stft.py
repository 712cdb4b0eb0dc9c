#!/usr/bin/env python3
"""Randomised fuzz of K_stft / K_istft against the numpy oracle (run on the GPU box): every power-of-two transform
size up to 8192 points plus (15 % of the cases) the 16384-point kernel and the four-step path up to 2^18 points, arbitrary
hops (1 .. 2 n_fft, odd ones included), zero-padding factors, window names, strided (channel-of-interleaved) input,
signals shorter than a frame; complex and magnitude modes; ISTFT round trips incl. explicit lengths."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from oracle import oracle_np as O
from pyaudiorestoration_amd import _lib, fourier as F

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t_end = time.time() + budget
case, worst = 0, 0.0


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))


while time.time() < t_end:
    rng = np.random.default_rng(seed0 + case)
    big = rng.random() < 0.15                          # the 16384-point kernel and the four-step path (up to 2^18 here)
    n_fft = int(2 ** rng.integers(13, 18)) if big else int(2 ** rng.integers(4, 14))
    zp = int(rng.choice([1, 1, 1, 2, 4]))
    while n_fft * zp > (1 << 18 if big else 8192):
        zp //= 2
    if big:
        hop = int(rng.choice([n_fft // 8, n_fft // 4, n_fft // 2, n_fft, n_fft + 7, 2 * n_fft, 5000, 4097]))
        n = int(rng.choice([n_fft // 2 + 2, n_fft, n_fft + 1, 3 * n_fft + 17, 5 * n_fft + 5]))
    else:
        hop = int(rng.choice([1, 3, n_fft // 8, n_fft // 4, n_fft // 2, n_fft, n_fft + 7, 2 * n_fft]))
        n = int(rng.choice([n_fft // 2 + 2, n_fft, n_fft + 1, 3 * n_fft + 17, 20 * n_fft + 5, 50000]))
    hop = max(1, hop)
    if hop <= 3:
        n = min(n, 4000)                               # keep the oracle's frame matrix small
    win = str(rng.choice(["hann", "blackmanharris", "hamming", "boxcar"]))
    ch = int(rng.choice([1, 2, 3]))
    inter = rng.standard_normal((n, ch)).astype(np.float32)
    x = inter[:, ch - 1]
    want = O.stft(x, n_fft, hop, win, zp)
    got = F.stft(x, n_fft, hop, win, zp)                              # strided numpy view in
    e1 = relerr(got, want)
    xt = torch.from_numpy(inter).cuda()[:, ch - 1]                    # strided device view in
    got_t = F.stft(xt, n_fft, hop, win, zp).cpu().numpy()
    e2 = relerr(got_t, want)
    mag = F.get_mag(xt, n_fft, hop, win, zp).cpu().numpy()
    e3 = relerr(mag, np.abs(want) + 1e-7)
    assert got.shape == want.shape == got_t.shape == mag.shape, (case, n_fft, hop, zp, n)
    errs = [e1, e2, e3]
    if zp == 1 and n_fft <= 8192:
        S = want.astype(np.complex64)
        for length in (n, None):
            y_want = O.istft(S, hop, win, length)
            y_got = F.istft(S, hop_length=hop, window_name=win, length=length)
            assert y_got.shape == y_want.shape, (case, "istft shape", n_fft, hop, n, length)
            # with hop > n_fft/2 the window-sumsquare gets arbitrarily small between frames and both float32
            # implementations divide rounding noise by it: only the well-conditioned overlaps are compared in value
            # float32 overlap-add of n_fft/hop terms per sample (both sides): rounding noise grows ~sqrt(overlap)
            if 2 * hop <= n_fft:
                errs.append(relerr(y_got, y_want) / max(1.0, (n_fft / hop / 16.0) ** 0.5))
    assert max(errs) < 1e-5, (case, errs, n_fft, hop, zp, n, win, ch)
    worst = max(worst, max(errs))
    case += 1
print(f"stft fuzz ok: {case} cases, worst relative error {worst:.2e}")
