#!/usr/bin/env python3
"""get_mag timings across transform sizes (single-workgroup kernels up to 8192, four-step above) on a 57.6 M-sample
resident signal, hop = n_fft / 4: Gsamples/s and algorithmic GB/s (4 B in + 4 (n_fft/2 + 1) / hop B out per sample)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.signal
import torch

from pyaudiorestoration_amd import _dev, _lib, fourier

n = 96000 * 600
x = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(_lib.lib().par_synth_signal_f32(0, _dev.ptr(x), 0, n, 96000.0, 0x5EED, _dev.stream_ptr(0)))
for n_fft in (32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 262144, 1048576):
    hop = n_fft // 4
    win = torch.from_numpy(scipy.signal.get_window("blackmanharris", n_fft).astype(np.float32)).cuda()
    fourier.stft_dev(x, n_fft, hop, win, 1, 1)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fourier.stft_dev(x, n_fft, hop, win, 1, 1)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    bytes_per_sample = 4 + 4 * (n_fft // 2 + 1) / hop
    print(f"n_fft {n_fft:8d} hop {hop:7d}: {best:8.3f} ms = {n / best / 1e6:7.1f} Gsamples/s = {n * bytes_per_sample / best / 1e6:7.0f} GB/s algorithmic")
