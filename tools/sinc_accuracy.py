#!/usr/bin/env python3
"""K_sinc (fused, NT = 32 and 50) against the C oracle on the signals that stress the tap-form error budgets: white
noise, a full-scale Nyquist tone (every approximation error adds with the same sign) and a full-scale tone at fs/4,
on a +-1 % speed curve: prints the error relative to the output peak."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import oracle_c as C
from pyaudiorestoration_amd import resampling as R

n = 400000
m = n // 256
st = np.linspace(0, n, m)
sp = 1.0 + 0.01 * np.sin(np.arange(m) * 0.05 + 0.3)
pos, _ = C.speed_to_pos(st, sp, n)
rng = np.random.default_rng(5)
t = np.arange(n)
signals = {"white noise": rng.standard_normal(n).astype(np.float32),
           "full-scale Nyquist tone": np.cos(np.pi * t).astype(np.float32),
           "full-scale tone at fs/4": np.cos(0.5 * np.pi * t + 0.1).astype(np.float32),
           "full-scale tone at 0.45 fs": np.cos(0.9 * np.pi * t + 0.2).astype(np.float32)}
plan = R.speed_plan_dev(torch.from_numpy(st).cuda(), torch.from_numpy(sp).cuda(), n, fused=True)
for NT in (32, 50):
    for name, sig in signals.items():
        want = C.sinc(pos, sig, NT, threads=16)
        got = R.varispeed_fused_dev(plan, torch.from_numpy(sig).cuda(), NT).cpu().numpy()
        via = R.sinc_resample_dev(torch.from_numpy(pos).cuda(), torch.from_numpy(sig).cuda(), NT).cpu().numpy()
        pk = np.max(np.abs(want))
        print(f"NT {NT:3d}  {name:28s} max |err| / peak = {np.max(np.abs(got - want)) / pk:.2e}   (position-array form "
              f"{np.max(np.abs(via - want)) / pk:.2e}, fused - position-array {np.max(np.abs(got - via)) / pk:.2e})")
