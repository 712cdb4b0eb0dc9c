#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU's share: the 512-file archive (192 kHz stereo, 10 min each) is dealt round-robin to
8 GPUs, so a GPU sees 64 files.  Files are synthesised in HBM (per-file seed = file index, curve phase 0.7 + index,
SURVEY 8d) one ahead of the resampler and go through resampling.varispeed_batch_dev: stereo K_sinc launch per file,
the next file's plan on a side stream underneath it.  Reports ms per file and channel-samples/s for the share; the
archive's wall time on 8 GPUs is this share's time (no communication)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pyaudiorestoration_amd import _dev, _lib, multi_gpu, resampling

files = int(sys.argv[1]) if len(sys.argv) > 1 else 64
world, rank = 8, 0
sr, seconds, NT = 192000, 600.0, 32
n, m = int(sr * seconds), int(seconds * sr / 256)
L = _lib.lib()
s = _dev.stream_ptr(0)
mine = multi_gpu.shard_items(512, world, rank)[:files]
bufs = [torch.empty((n, 2), dtype=torch.float32, device="cuda") for _ in range(3)]      # ring of resident files
mono = torch.empty(n, dtype=torch.float32, device="cuda")
curves = [(torch.empty(m, dtype=torch.float64, device="cuda"), torch.empty(m, dtype=torch.float64, device="cuda")) for _ in range(3)]


def produce():
    for k, idx in enumerate(mine):
        sig, (st, sp) = bufs[k % 3], curves[k % 3]
        for c in range(2):
            _lib.check(L.par_synth_signal_f32(0, _dev.ptr(mono), 0, n, float(sr), 2 * idx + c, s))
            sig[:, c] = mono
        _lib.check(L.par_synth_speed_curve_f64(0, _dev.ptr(st), _dev.ptr(sp), m, seconds, float(sr), 0.01, 0.55, 0.7 + idx, s))
        yield st, sp, sig


total = 0
paths = []
for _ in resampling.varispeed_batch_dev(list(produce())[:3], NT):      # warm-up: allocator pools, first launches
    pass
torch.cuda.synchronize()
t0 = time.perf_counter()
for k, out, plan in resampling.varispeed_batch_dev(produce(), NT):
    total += 2 * plan.len_out
    paths.append(plan.path)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
# the synthetic generator is part of the loop here (it stands in for the H2D upload of a real archive): time it alone
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in produce():
    pass
torch.cuda.synchronize()
gen = time.perf_counter() - t1
print(json.dumps({"workload": f"config 5, one GPU's share: {len(mine)} of 512 files, 600 s @192 kHz stereo",
                  "ms_per_file_incl_generation": round(dt / len(mine) * 1e3, 3), "ms_per_file_generation_only": round(gen / len(mine) * 1e3, 3),
                  "ms_per_file_resample": round((dt - gen) / len(mine) * 1e3, 3),
                  "channel_Gsamples_per_s_resample": round(total / (dt - gen) / 1e9, 1),
                  "archive_wall_s_on_8_gpus_resample": round((dt - gen), 3),
                  "plans_on_serial_host_path": int(sum(paths))}, indent=1))
