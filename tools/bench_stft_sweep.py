#!/usr/bin/env python3
"""K_stft across hops, zero-padding factors and output forms on a 23 M-sample resident signal: ms and the algorithmic
GB/s (4 B in per sample + the bins written), to spot sizes that fall off the curve."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.signal
import torch

from pyaudiorestoration_amd import fourier

n = 96000 * 240
x = torch.randn(n, dtype=torch.float32, device="cuda")
for n_fft in (256, 512, 1024, 2048, 4096, 8192):
    win = torch.from_numpy(scipy.signal.get_window("hann", n_fft).astype(np.float32)).cuda()
    for zp in (1, 4):
        if n_fft * zp > 16384:
            continue
        for div in (1, 2, 4, 8, 16):
            hop = n_fft // div
            for mode in (1, 0):
                bins = n_fft * zp // 2 + 1
                frames = n // hop + 1
                if frames * bins * (4 if mode else 8) > 12e9:
                    continue
                fourier.stft_dev(x, n_fft, hop, win, zp, mode)
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fourier.stft_dev(x, n_fft, hop, win, zp, mode)
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1))
                gb = (4 * n + frames * bins * (4 if mode else 8)) / 1e9
                print(f"stft n_fft {n_fft:5d} zp {zp} hop n/{div:<2d} {'mag' if mode else 'c64'}: {best:7.3f} ms, {gb / best * 1e3:6.0f} GB/s algorithmic, "
                      f"{frames * 1e-3 / best:7.1f} Mframes/s")
