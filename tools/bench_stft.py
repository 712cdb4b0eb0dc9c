#!/usr/bin/env python3
"""Secondary measurement: K_stft magnitude (get_mag) throughput vs its HBM roofline (12.02 B per input
sample at 1024/256) and vs the reference's own GPU route torch.stft (util/fourier.py:101-107)."""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.signal
import torch

from pyaudiorestoration_amd import _dev, _lib, fourier

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 96000 * 600
n_fft, hop = 1024, 256
dev = 0
x = torch.empty(n, dtype=torch.float32, device="cuda")
L = _lib.lib()
_lib.check(L.par_synth_signal_f32(dev, _dev.ptr(x), 0, n, 96000.0, 0x5EED, _dev.stream_ptr(dev)))
win = torch.from_numpy(scipy.signal.get_window("blackmanharris", n_fft).astype(np.float32)).cuda()


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


res = {}
for mode, name, bps in ((1, "get_mag", 4 + 4 * (n_fft / 2 + 1) / hop), (0, "stft_complex", 4 + 8 * (n_fft / 2 + 1) / hop)):
    t = timeit(lambda: fourier.stft_dev(x, n_fft, hop, win, 1, mode))
    res[name] = {"ms": t * 1e3, "Msamples/s": n / t / 1e6, "GB/s_algorithmic": n * bps / t / 1e9, "frac_of_8TB/s": n * bps / t / 8e12}


def ref_torch():
    r = torch.stft(x, n_fft, hop_length=hop, window=win, win_length=n_fft, center=True, pad_mode="reflect", normalized=False,
                   onesided=True, return_complex=True)
    r /= math.sqrt(n_fft)
    return r.abs() + 1e-7


t = timeit(ref_torch, 5)
res["torch.stft+abs (reference GPU route, no D2H)"] = {"ms": t * 1e3, "Msamples/s": n / t / 1e6}
print(json.dumps({"n": n, "n_fft": n_fft, "hop": hop, **res}, indent=1))
