#!/usr/bin/env python3
"""BASELINE config 4 measurement: dropout inpaint on a 322 531-sample signal tiled x256 (82.6 M samples),
STFT 512/32 blackmanharris -> 32 markers per tile (8192 boxes, ONE K_heal launch) -> gain -> ISTFT, all
resident in HBM.  Algorithmic bytes: 136.5 B per input sample (materialised c64 spectrogram
written and re-read).  Synthetic stand-in for dropouts_sample.flac with the same length and rate."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.signal
import torch

from pyaudiorestoration_amd import _dev, _lib, fourier, pipeline

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n1, sr, n_fft, hop = 322531, 44100, 512, 32
dev = 0
L = _lib.lib()
n = n1 * tiles
x = torch.zeros(n + n_fft // 2, dtype=torch.float32, device="cuda")          # fix_length(n + fft/2)
_lib.check(L.par_synth_signal_f32(dev, _dev.ptr(x), 0, n, float(sr), 0x5EED, _dev.stream_ptr(dev)))
rng = np.random.default_rng(4)
marks = []
for k in range(tiles):
    for t in np.sort(rng.uniform(0.2, n1 / sr - 0.2, 32)):
        w = rng.uniform(0.004, 0.02)
        marks.append((k * n1 / sr + t - w / 2, 500.0, k * n1 / sr + t + w / 2, 9000.0, 0.5))
geo = np.array([pipeline.marker_geometry(m, sr, hop, n_fft) for m in marks], dtype=np.int32)
geo_t = torch.from_numpy(geo).cuda()
win = torch.from_numpy(scipy.signal.get_window("blackmanharris", n_fft).astype(np.float32)).cuda()
frames = int(L.par_stft_frames(n + n_fft // 2, n_fft, hop))
bins = n_fft // 2 + 1
spec = torch.empty((frames, bins), dtype=torch.complex64, device="cuda")
gain = torch.zeros((frames, bins), dtype=torch.float32, device="cuda")      # zero-filled ONCE; apply-and-clear keeps it zero
assert L.par_istft_scratch_floats(frames, n_fft, hop) == 0             # frames are overlap-added in LDS
y = torch.empty(n, dtype=torch.float32, device="cuda")
s = _dev.stream_ptr(dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]


def step(rec=False):
    if rec: ev[0].record()
    _lib.check(L.par_stft_f32(dev, _dev.ptr(x), n + n_fft // 2, 1, n_fft, hop, 1, _dev.ptr(win), _dev.ptr(spec), 0, 0, s))
    if rec: ev[1].record()
    _lib.check(L.par_inpaint_gain_db_c64(dev, _dev.ptr(spec), frames, bins, _dev.ptr(geo_t), len(geo), _dev.ptr(gain), s))
    if rec: ev[2].record()
    _lib.check(L.par_spec_apply_gain_boxes_c64(dev, _dev.ptr(spec), frames, bins, _dev.ptr(geo_t), len(geo), _dev.ptr(gain), s))
    if rec: ev[3].record()
    _lib.check(L.par_istft_f32(dev, _dev.ptr(spec), frames, n_fft, hop, _dev.ptr(win), None, _dev.ptr(y), n, n_fft // 2, s))
    if rec: ev[4].record()


for _ in range(2):
    step()
torch.cuda.synchronize()
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
step(True)
torch.cuda.synchronize()
names = ("stft_c64", "inpaint_gain", "apply_gain_boxes", "istft")
parts = {nm: ev[i].elapsed_time(ev[i + 1]) for i, nm in enumerate(names)}
print(json.dumps({"workload": f"config4 x{tiles}", "samples": n, "frames": frames, "markers": len(geo), "ms": dt * 1e3,
                  "Msamples/s": n / dt / 1e6, "GB/s_algorithmic(136.5B/sample)": n * 136.5 / dt / 1e9,
                  "frac_of_8TB/s": n * 136.5 / dt / 8e12, "parts_ms": parts, "hbm_GiB": (spec.numel() * 8 + gain.numel() * 4) / 2**30, "mask_is_zero_again": bool((gain == 0).all()),
                  "healed_rms": float(y.float().pow(2).mean().sqrt())}, indent=1))
