#!/usr/bin/env python3
"""How the plan scales with the DENSITY of the speed curve (same 10-min 192 kHz file, same overall speed ramp): the
reference's per-segment cumsum is a strictly sequential float64 chain, evaluated here by one lane per segment, so
few long segments mean little parallelism in the planning pass (K_sinc itself is unaffected: it restarts from the
cumsum checkpoints every 8 samples)."""
import json
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
res = []
for m in (2, 11, 601, 60001, int(seconds * sr / 256)):
    st = np.linspace(0, n, m)
    sp = np.linspace(0.995, 1.0051, m)
    st_t, sp_t = torch.from_numpy(st).cuda(), torch.from_numpy(sp).cuda()
    plan = R.speed_plan_dev(st_t, sp_t, n, fused=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan = R.speed_plan_dev(st_t, sp_t, n, fused=True)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t0
    out = R.varispeed_fused_dev(plan, sig, 32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = R.varispeed_fused_dev(plan, sig, 32)
    torch.cuda.synchronize()
    res.append({"curve_points": m, "samples_per_segment": n // max(m - 1, 1), "plan_ms": round(t_plan * 1e3, 3),
                "k_sinc_ms": round((time.perf_counter() - t0) * 1e3, 3), "fused_ok": plan.fused_ok, "path": plan.path})
print(json.dumps(res, indent=1))
