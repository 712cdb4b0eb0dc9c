#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer boundaries (never bench.py's `value`): numpy in -> numpy out through
resampling.run's device flow (signal H2D, plan + fused K_sinc, output D2H) and through the operator slot
sinc_wrapper (which also ships a float64 sample_at array over PCIe)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import inputs
from pyaudiorestoration_amd import _dev, resampling

sr, seconds = 192000, float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
n = int(sr * seconds)
sig = inputs.bench_signal(0, n, sr)
curve = inputs.bench_speed_curve(seconds, sr)
res = {"samples": n}


def wall(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps, out


def flow_pageable(staged=True):
    sig_t = _dev.to_dev(sig, torch.float32, 0) if staged else torch.from_numpy(sig).to("cuda:0")
    st_t = _dev.to_dev(curve[:, 0] * sr, torch.float64, 0)
    sp_t = _dev.to_dev(np.ascontiguousarray(curve[:, 1]), torch.float64, 0)
    plan = resampling.speed_plan_dev(st_t, sp_t, n, fused=True)
    out = resampling.varispeed_fused_dev(plan, sig_t, 32)
    return _dev.to_host(out) if staged else out.cpu().numpy()


dt, y = wall(flow_pageable)
res["run()-style flow, pageable numpy in/out (r06: through the pinned chunk ring, _dev.to_dev / to_host)"] = {"s": round(dt, 4), "Msamples/s": round(len(y) / dt / 1e6, 1)}
dt, y0 = wall(lambda: flow_pageable(False))
res["the same with the runtime's own pageable copies (r05)"] = {"s": round(dt, 4), "Msamples/s": round(len(y0) / dt / 1e6, 1)}
assert np.array_equal(y, y0)

pin_in = torch.from_numpy(sig).pin_memory()
pin_out = torch.empty(int(n * 1.02) + 1024, dtype=torch.float32).pin_memory()


def flow_pinned():
    sig_t = pin_in.to("cuda:0", non_blocking=True)
    st_t = _dev.to_dev(curve[:, 0] * sr, torch.float64, 0)
    sp_t = _dev.to_dev(np.ascontiguousarray(curve[:, 1]), torch.float64, 0)
    plan = resampling.speed_plan_dev(st_t, sp_t, n, fused=True)
    out = resampling.varispeed_fused_dev(plan, sig_t, 32)
    pin_out[:out.numel()].copy_(out, non_blocking=True)
    torch.cuda.synchronize()
    return pin_out[:out.numel()]


dt, y2 = wall(flow_pinned)
res["same flow, pinned host buffers"] = {"s": round(dt, 4), "Msamples/s": round(len(y2) / dt / 1e6, 1)}
pos = resampling.speed_to_pos(curve[:, 0] * sr, curve[:, 1], n)
dt, y3 = wall(lambda: resampling.sinc_wrapper(pos, sig, 0, 32), 2)
res["sinc_wrapper(sample_at f64, signal) operator slot"] = {"s": round(dt, 4), "Msamples/s": round(len(y3) / dt / 1e6, 1)}
assert np.array_equal(y, y2.numpy())

# operator slot #1 at scale: numpy signal in, numpy magnitude spectrogram out (1024 / 256: 8 B down per sample in)
from pyaudiorestoration_amd import fourier
dt, m1 = wall(lambda: fourier.get_mag(sig, 1024, 256, "blackmanharris"), 2)
res["get_mag(numpy signal) -> numpy magnitudes, 1024/256 (operator slot #1; staged transfers)"] = {"s": round(dt, 4), "Msamples/s": round(n / dt / 1e6, 1)}
keep = _dev._STAGE_MIN
_dev._STAGE_MIN = 1 << 62
dt, m0 = wall(lambda: fourier.get_mag(sig, 1024, 256, "blackmanharris"), 2)
_dev._STAGE_MIN = keep
res["the same with the runtime's own pageable copies (r05)"  + " "] = {"s": round(dt, 4), "Msamples/s": round(n / dt / 1e6, 1)}
assert np.array_equal(m1, m0)
del m1, m0

# a batch of host files through varispeed_batch_host: upload of file k+1 under the download of file k
n_files = 12
st_np, sp_np = curve[:, 0] * sr, np.ascontiguousarray(curve[:, 1])
for label, src in (("pinned inputs", pin_in), ("pageable numpy inputs (staged)", sig)):
    def batch():
        total = 0
        for k, out in resampling.varispeed_batch_host(((st_np, sp_np, src) for _ in range(n_files)), 32):
            total += out.shape[0]
        return total
    batch()
    t0 = time.perf_counter()
    total = batch()
    dt = time.perf_counter() - t0
    res[f"varispeed_batch_host, {n_files} files, {label}"] = {"s_per_file": round(dt / n_files, 4),
                                                            "Msamples/s": round(total / dt / 1e6, 1)}
stereo = torch.stack((pin_in, pin_in.flip(0)), dim=1).contiguous().pin_memory()


def batch2():
    total = 0
    for k, out in resampling.varispeed_batch_host(((st_np, sp_np, stereo) for _ in range(n_files)), 32):
        total += out.shape[0] * 2
    return total


batch2()
t0 = time.perf_counter()
total = batch2()
dt = time.perf_counter() - t0
res[f"varispeed_batch_host, {n_files} stereo files, pinned"] = {"s_per_file": round(dt / n_files, 4),
                                                               "M channel-samples/s": round(total / dt / 1e6, 1)}
print(json.dumps(res, indent=1))
