"""Flag statistics of a fused plan's block records on the benchmark curve (how many outputs leave the record model):
    python tools/rec_stats.py [seconds]
Decodes the aux buffer on the host (layout: csrc/pos_plan.h fused_aux_view)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pyaudiorestoration_amd import _dev, _lib

L = _lib.lib()
dev, sr, seconds = 0, 192000, float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
s = _dev.stream_ptr(dev)
n = int(sr * seconds)
m = int(seconds * sr / 256)
st = torch.empty(m, dtype=torch.float64, device="cuda")
sp = torch.empty(m, dtype=torch.float64, device="cuda")
_lib.check(L.par_synth_speed_curve_f64(dev, _dev.ptr(st), _dev.ptr(sp), m, seconds, float(sr), 0.01, 0.55, 0.7, s))
cap = int(n * 1.02) + 1024
nb, ab = int(L.par_speed_plan_bytes(m)), int(L.par_fused_aux_bytes(cap, m))
work = torch.empty(nb, dtype=torch.uint8, device="cuda")
aux = torch.zeros(ab, dtype=torch.uint8, device="cuda")
lo, tr, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
_lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st), _dev.ptr(sp), m, n, _dev.ptr(work), nb, _dev.ptr(aux), ab, cap,
                                         ctypes.byref(lo), ctypes.byref(tr), 0, None, ctypes.byref(ok), s))
torch.cuda.synchronize()
a = aux.cpu().numpy()
ck_len, tiles = cap // 8 + m + 16, cap // 1024 + 4
blocks = tiles * 32                                     # kRec = 32 outputs per record
off = (ck_len + tiles) * 8 + m * 32 + tiles * 32
rec = a[off:off + blocks * 16].view(np.uint32).reshape(blocks, 4)
nblk = (lo.value + 31) // 32
w = rec[:nblk, 0]
ustar = (w & 31) + 1
e0, e1, slow0, slow1, cubic = (w >> 5) & 1, (w >> 6) & 1, (w >> 7) & 1, (w >> 8) & 1, (w >> 9) & 1
boundary = ustar < 32
print(f"{lo.value} outputs, {nblk} blocks of 32, ok={ok.value}")
print(f"boundary blocks {boundary.sum()} ({boundary.mean():.4%}), slow0 {slow0.sum()}, slow1 among boundary {slow1[boundary].sum()} "
      f"({slow1[boundary].mean():.2%}), E0 {e0.sum()}, E1 {e1.sum()}, cubic {cubic.sum()}")
wv = nblk // 8                                          # waves of 256 outputs = 8 blocks
anyslow = ((slow0[:wv * 8] | (slow1[:wv * 8] & boundary[:wv * 8])).reshape(wv, 8).max(axis=1))
print(f"waves with a slow piece: {anyslow.sum()} of {wv} ({anyslow.mean():.2%})")
