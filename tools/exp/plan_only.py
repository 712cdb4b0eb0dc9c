"""Twenty plans of the benchmark file back to back, nothing else (for rocprofv3 --kernel-trace: the plan's kernels alone)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pyaudiorestoration_amd import _dev, _lib
L = _lib.lib()
dev, sr, seconds = 0, 192000, float(os.environ.get("SECONDS_", "3600"))
s = _dev.stream_ptr(dev)
n = int(sr * seconds); m = int(seconds * sr / 256)
st = torch.empty(m, dtype=torch.float64, device="cuda"); sp = torch.empty(m, dtype=torch.float64, device="cuda")
_lib.check(L.par_synth_speed_curve_f64(dev, _dev.ptr(st), _dev.ptr(sp), m, seconds, float(sr), 0.01, 0.55, 0.7, s))
cap = int(n * 1.02) + 1024
nb, ab = int(L.par_speed_plan_bytes(m)), int(L.par_fused_aux_bytes(cap, m))
work = torch.empty(nb, dtype=torch.uint8, device="cuda"); aux = torch.empty(ab, dtype=torch.uint8, device="cuda")
lo, tr, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
def plan():
    _lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st), _dev.ptr(sp), m, n, _dev.ptr(work), nb, _dev.ptr(aux), ab, cap,
                                             ctypes.byref(lo), ctypes.byref(tr), 0, None, ctypes.byref(ok), s))
for _ in range(3): plan()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): plan()
torch.cuda.synchronize(); print(f"plan alone {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
