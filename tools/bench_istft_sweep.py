#!/usr/bin/env python3
"""ISTFT across transform sizes and overlaps on ~23 M output samples (device spectrogram in, device signal out): the
fused kernel (overlap-add in LDS, s - 1 = n_fft/hop - 1 frames of every workgroup re-transformed) against the two-kernel
form (every frame transformed once, frames through HBM).  PAR_HIP_LIB selects the build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.signal
import torch

from pyaudiorestoration_amd import fourier

n = 96000 * 240
for n_fft in (256, 512, 1024, 2048, 4096, 8192):
    for div in (2, 4, 8, 16):
        hop = n_fft // div
        frames = n // hop + 1
        S = torch.view_as_complex(torch.randn(frames, n_fft // 2 + 1, 2, dtype=torch.float32, device="cuda")).T
        win = torch.from_numpy(scipy.signal.get_window("hann", n_fft).astype(np.float32)).cuda()
        fourier.istft_dev(S, hop, win, length=n)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fourier.istft_dev(S, hop, win, length=n)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(f"istft n_fft {n_fft:5d} hop n/{div:<2d}: {best:7.3f} ms = {n / best / 1e6:6.1f} Gsamples/s")
        del S
