"""Is K_sinc held by the board's power limit?  The same launch (60-min file, benchmark curve, back to back for ~2 s) on the benchmark's
synthetic signal, on white noise and on an all-zero signal: identical instruction streams, different switching activity."""
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
st = torch.empty(m, dtype=torch.float64, device="cuda"); sp = torch.empty(m, dtype=torch.float64, device="cuda")
_lib.check(L.par_synth_speed_curve_f64(dev, _dev.ptr(st), _dev.ptr(sp), m, seconds, float(sr), 0.01, 0.55, 0.7, s))
cap = int(n * 1.02) + 1024
nb, ab = int(L.par_speed_plan_bytes(m)), int(L.par_fused_aux_bytes(cap, m))
work = torch.empty(nb, dtype=torch.uint8, device="cuda"); aux = torch.empty(ab, dtype=torch.uint8, device="cuda")
out = torch.empty(cap, dtype=torch.float32, device="cuda")
lo, tr, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
_lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st), _dev.ptr(sp), m, n, _dev.ptr(work), nb, _dev.ptr(aux), ab, cap,
                                         ctypes.byref(lo), ctypes.byref(tr), 0, None, ctypes.byref(ok), s))
def run():
    _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(sp), m, _dev.ptr(work), _dev.ptr(aux), cap, lo.value, _dev.ptr(sig), 1, n, nt, _dev.ptr(out), 1, s))
def fill(kind):
    if kind == "benchmark signal": _lib.check(L.par_synth_signal_f32(dev, _dev.ptr(sig), 0, n, float(sr), 0x5EED, s))
    elif kind == "white noise": sig.normal_()
    elif kind == "zeros": sig.zero_()
    elif kind == "constant 0.5": sig.fill_(0.5)
for rep in range(2):
    for kind in ("benchmark signal", "white noise", "zeros", "constant 0.5"):
        fill(kind)
        for _ in range(100): run()                  # ~0.5 s to settle the clocks
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): run()
        e1.record(); torch.cuda.synchronize()
        print(f"{kind:18s} {e0.elapsed_time(e1) / 200:.3f} ms per launch", flush=True)
