#!/usr/bin/env python3
"""K_sosfiltfilt across signal lengths and filter orders (device float64 signal in and out; the Butterworth design
on the host is included: it is part of every call)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pyaudiorestoration_amd import filters

for order in (1, 3, 5, 8):
    for n in (10_000, 100_000, 1_000_000, 4_000_000, 16_000_000, 33_554_432, 40_000_000, 100_000_000):
        x = torch.randn(n, dtype=torch.float64, device="cuda")
        filters.bandpass_dev(x, 300.0, 6000.0, 48000.0, order)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            filters.bandpass_dev(x, 300.0, 6000.0, 48000.0, order)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"bandpass order {order} n {n:10d}: {best * 1e3:8.3f} ms = {n / best / 1e9:6.2f} Gsamples/s")
        del x
