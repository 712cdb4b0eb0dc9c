#!/usr/bin/env python3
"""get_mag (K_stft mode 1, 1024/256) three times on 57.6 M samples -- for rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes."""
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
win = torch.from_numpy(scipy.signal.get_window("blackmanharris", 1024).astype(np.float32)).cuda()
for _ in range(3):
    fourier.stft_dev(x, 1024, 256, win, 1, 1)
torch.cuda.synchronize()
print("frames", n // 256 + 1, "algorithmic magnitude bytes", (n // 256 + 1) * 513 * 4)
if "--time" in sys.argv:
    best = 1e9
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fourier.stft_dev(x, 1024, 256, win, 1, 1)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"get_mag 1024/256 on {n} samples: {best:.4f} ms = {n / best / 1e6:.1f} Gsamples/s")
