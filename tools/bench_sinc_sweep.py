#!/usr/bin/env python3
"""K_sinc across its parameter space on one 10-min 192 kHz file (plan excluded): tap counts NT (32 and 50 have
specialised fully unrolled tap loops, the others run the generic loop), speed ranges (below 1: fc = 1, the cheap tap
form; above 1: fc < 1 on every output; around 1: the bench's mix), mono and the two stereo layouts (interleaved = the
hot path with LDS-DMA staging, planar = the general path).  Prints ms, G output samples/s and ns per tap-output."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pyaudiorestoration_amd import resampling as R

sr, seconds = 192000, float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
n = int(sr * seconds)
sig = torch.randn(n, dtype=torch.float32, device="cuda")
st2 = torch.randn(n, 2, dtype=torch.float32, device="cuda")
pl0, pl1 = st2[:, 0].contiguous(), st2[:, 1].contiguous()
m = n // 256


def timed(f):
    f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


for lo, hi, tag in ((0.99, 1.01, "speed 0.99..1.01"), (0.90, 0.999, "speed 0.90..0.999 (fc = 1)"), (1.001, 1.10, "speed 1.001..1.10 (fc < 1)"),
                    (0.5, 2.0, "speed 0.5..2")):
    t = np.linspace(0, n, m)
    sp = lo + (hi - lo) * (0.5 + 0.5 * np.sin(2 * np.pi * 0.55 * t / sr))
    plan = R.speed_plan_dev(torch.from_numpy(t).cuda(), torch.from_numpy(sp).cuda(), n, fused=True)
    n_out = int(plan.len_out)
    for nt in (4, 16, 32, 50, 64, 100):
        dt = timed(lambda: R.varispeed_fused_dev(plan, sig, nt))
        print(f"{tag:32s} mono   NT {nt:3d}: {dt * 1e3:8.3f} ms = {n_out / dt / 1e9:7.2f} Gs/s, {dt / n_out / (2 * nt) * 1e12:6.2f} ps per tap")
    if lo == 0.99:
        for nt in (32, 50):
            o = torch.empty(n_out, 2, dtype=torch.float32, device="cuda")
            dt = timed(lambda: R.varispeed_fused_stereo_dev(plan, st2[:, 0], st2[:, 1], nt, o[:, 0], o[:, 1], sig_stride=2, len_in=n, out_stride=2))
            print(f"{tag:32s} stereo interleaved NT {nt:3d}: {dt * 1e3:8.3f} ms = {2 * n_out / dt / 1e9:7.2f} G ch-samples/s")
            o0, o1 = torch.empty(n_out, dtype=torch.float32, device="cuda"), torch.empty(n_out, dtype=torch.float32, device="cuda")
            dt = timed(lambda: R.varispeed_fused_stereo_dev(plan, pl0, pl1, nt, o0, o1))
            print(f"{tag:32s} stereo planar      NT {nt:3d}: {dt * 1e3:8.3f} ms = {2 * n_out / dt / 1e9:7.2f} G ch-samples/s")
