"""What the plan costs K_sinc when it runs beside it (side stream): the real plan, a plan of a 1000-point curve (the same
launch sequence, no work), no plan.  K_sinc of the 60-min benchmark file back to back on the main stream."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pyaudiorestoration_amd import _dev, _lib
L = _lib.lib()
dev, sr, seconds, nt = 0, 192000, 3600.0, 32
s = _dev.stream_ptr(dev)
side = torch.cuda.Stream(device=dev)
sp_side = ctypes.c_void_p(side.cuda_stream)
def curve(seconds_, m_):
    st = torch.empty(m_, dtype=torch.float64, device="cuda"); sp = torch.empty(m_, dtype=torch.float64, device="cuda")
    _lib.check(L.par_synth_speed_curve_f64(dev, _dev.ptr(st), _dev.ptr(sp), m_, seconds_, float(sr), 0.01, 0.55, 0.7, s))
    return st, sp
n = int(sr * seconds); m = int(seconds * sr / 256)
sig = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(L.par_synth_signal_f32(dev, _dev.ptr(sig), 0, n, float(sr), 0x5EED, s))
st, sp = curve(seconds, m)
cap = int(n * 1.02) + 1024
def bufs(m_, cap_):
    nb, ab = int(L.par_speed_plan_bytes(m_)), int(L.par_fused_aux_bytes(cap_, m_))
    return torch.empty(nb, dtype=torch.uint8, device="cuda"), torch.empty(ab, dtype=torch.uint8, device="cuda"), nb, ab
work, aux, nb, ab = bufs(m, cap)
work2, aux2, nb2, ab2 = bufs(m, cap)
out = torch.empty(cap, dtype=torch.float32, device="cuda")
lo, tr, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
_lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st), _dev.ptr(sp), m, n, _dev.ptr(work), nb, _dev.ptr(aux), ab, cap,
                                         ctypes.byref(lo), ctypes.byref(tr), 0, None, ctypes.byref(ok), s))
len_out = lo.value
# small curve: 1000 points over 1.3 s
ms_, ns_ = 1000, 256000
sts, sps = curve(ns_ / sr, ms_)
caps = int(ns_ * 1.02) + 1024
works, auxs, nbs_, abs_ = bufs(ms_, caps)
def sinc():
    _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(sp), m, _dev.ptr(work), _dev.ptr(aux), cap, len_out, _dev.ptr(sig), 1, n, nt, _dev.ptr(out), 1, s))
def plan_real():
    l2 = ctypes.c_int64(0)
    _lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st), _dev.ptr(sp), m, n, _dev.ptr(work2), nb2, _dev.ptr(aux2), ab2, cap,
                                             ctypes.byref(l2), ctypes.byref(tr), 0, None, ctypes.byref(ok), sp_side))
def plan_small():
    l2 = ctypes.c_int64(0)
    _lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(sts), _dev.ptr(sps), ms_, ns_, _dev.ptr(works), nbs_, _dev.ptr(auxs), abs_, caps,
                                             ctypes.byref(l2), ctypes.byref(tr), 0, None, ctypes.byref(ok), sp_side))
for name, pl in (("no plan", None), ("small plan (launches only)", plan_small), ("real plan", plan_real), ("no plan", None)):
    for _ in range(3):
        sinc()
        if pl: pl()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        sinc()
        if pl: pl()
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step")
