"""The plan of the benchmark file ALONE (no K_sinc beside it): ms per plan, 20 plans back to back (each ends in its header
read-back).  And plan + K_sinc strictly one after the other on one stream."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pyaudiorestoration_amd import _dev, _lib
L = _lib.lib()
dev, sr, seconds, nt = 0, 192000, 3600.0, 32
s = _dev.stream_ptr(dev)
n = int(sr * seconds); m = int(seconds * sr / 256)
sig = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(L.par_synth_signal_f32(dev, _dev.ptr(sig), 0, n, float(sr), 0x5EED, s))
st = torch.empty(m, dtype=torch.float64, device="cuda"); sp = torch.empty(m, dtype=torch.float64, device="cuda")
_lib.check(L.par_synth_speed_curve_f64(dev, _dev.ptr(st), _dev.ptr(sp), m, seconds, float(sr), 0.01, 0.55, 0.7, s))
cap = int(n * 1.02) + 1024
nb, ab = int(L.par_speed_plan_bytes(m)), int(L.par_fused_aux_bytes(cap, m))
work = torch.empty(nb, dtype=torch.uint8, device="cuda"); aux = torch.empty(ab, dtype=torch.uint8, device="cuda")
out = torch.empty(cap, dtype=torch.float32, device="cuda")
lo, tr, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
def plan():
    _lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st), _dev.ptr(sp), m, n, _dev.ptr(work), nb, _dev.ptr(aux), ab, cap,
                                             ctypes.byref(lo), ctypes.byref(tr), 0, None, ctypes.byref(ok), s))
def sinc():
    _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(sp), m, _dev.ptr(work), _dev.ptr(aux), cap, lo.value, _dev.ptr(sig), 1, n, nt, _dev.ptr(out), 1, s))
for _ in range(3): plan(); sinc()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): plan()
torch.cuda.synchronize(); tp = (time.perf_counter() - t0) / 20
t0 = time.perf_counter()
for _ in range(20): sinc()
torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / 20
t0 = time.perf_counter()
for _ in range(20): plan(); sinc()
torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / 20
print(f"plan alone {tp*1e3:.3f} ms   K_sinc alone {ts*1e3:.3f} ms   plan then K_sinc (one stream, serial) {tb*1e3:.3f} ms")
