"""K_sinc launches of 10-min stereo files back to back on ONE stream against alternating over TWO streams (does the head of one
launch fill the tail of the other?).  Each stream has its own plan / aux (the tile list lives in aux)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pyaudiorestoration_amd import _dev, _lib
L = _lib.lib()
dev, sr, seconds = 0, 192000, float(os.environ.get("SECONDS_", "600"))
mono_mode = bool(os.environ.get("MONO"))
n = int(sr * seconds); m = int(seconds * sr / 256)
s0 = _dev.stream_ptr(dev)
mono = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(L.par_synth_signal_f32(dev, _dev.ptr(mono), 0, n, float(sr), 0x5EED, s0))
sig = mono if mono_mode else torch.stack((mono, mono.flip(0)), dim=1).contiguous().reshape(-1)
st = torch.empty(m, dtype=torch.float64, device="cuda"); sp = torch.empty(m, dtype=torch.float64, device="cuda")
_lib.check(L.par_synth_speed_curve_f64(dev, _dev.ptr(st), _dev.ptr(sp), m, seconds, float(sr), 0.01, 0.55, 0.7, s0))
cap = int(n * 1.02) + 1024
nb, ab = int(L.par_speed_plan_bytes(m)), int(L.par_fused_aux_bytes(cap, m))
slots = []
for k in range(2):
    work = torch.empty(nb, dtype=torch.uint8, device="cuda"); aux = torch.empty(ab, dtype=torch.uint8, device="cuda")
    out = torch.empty(cap * (1 if mono_mode else 2), dtype=torch.float32, device="cuda")
    lo, tr, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st), _dev.ptr(sp), m, n, _dev.ptr(work), nb, _dev.ptr(aux), ab, cap,
                                             ctypes.byref(lo), ctypes.byref(tr), 0, None, ctypes.byref(ok), s0))
    slots.append((work, aux, out, lo.value))
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def launch(k, stream):
    work, aux, out, lo = slots[k]
    sp_ = stream.cuda_stream
    if mono_mode:
        _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(sp), m, _dev.ptr(work), _dev.ptr(aux), cap, lo, _dev.ptr(sig), 1, n, 32, _dev.ptr(out), 1, ctypes.c_void_p(sp_)))
    else:
        _lib.check(L.par_varispeed_fused_stereo_f32(dev, _dev.ptr(sp), m, _dev.ptr(work), _dev.ptr(aux), cap, lo, _dev.ptr(sig), ctypes.c_void_p(sig.data_ptr() + 4), 2, n, 32,
                                                    _dev.ptr(out), ctypes.c_void_p(out.data_ptr() + 4), 2, ctypes.c_void_p(sp_)))
import time
for name, pick in (("one stream ", lambda i: 0), ("two streams", lambda i: i & 1), ("one stream ", lambda i: 0), ("two streams", lambda i: i & 1)):
    for i in range(4): launch(i & 1, streams[pick(i)])
    torch.cuda.synchronize()
    N = 40
    t0 = time.perf_counter()
    for i in range(N): launch(i & 1, streams[pick(i)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print(f"{name}: {dt * 1e3:.3f} ms per file  {(1 if mono_mode else 2) * slots[0][3] / dt / 1e9:.1f} G", flush=True)
