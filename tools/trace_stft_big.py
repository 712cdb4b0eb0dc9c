#!/usr/bin/env python3
"""Three get_mag calls each at 65536 and 2^20 points (hop = n_fft / 4) on a 57.6 M-sample resident signal, for
rocprofv3 --kernel-trace --stats: which of the four-step path's kernels (column pass, row pass, untangle) takes what."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.signal, torch
from pyaudiorestoration_amd import _dev, _lib, fourier
n = 96000 * 600
x = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(_lib.lib().par_synth_signal_f32(0, _dev.ptr(x), 0, n, 96000.0, 0x5EED, _dev.stream_ptr(0)))
for n_fft in (65536, 1048576):
    win = torch.from_numpy(scipy.signal.get_window("blackmanharris", n_fft).astype(np.float32)).cuda()
    for _ in range(3):
        fourier.stft_dev(x, n_fft, n_fft // 4, win, 1, 1)
torch.cuda.synchronize()
