"""K_sinc (10-min mono file, benchmark curve) alone and with a side stream launching tiny workgroups: does workgroup DISPATCH on
another queue cost K_sinc time?  The plan of a 60-min file is ~350k small workgroups."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pyaudiorestoration_amd import _dev, _lib
L = _lib.lib()
E = ctypes.CDLL(os.path.join(ROOT, "tools", "exp", "libempty_wgs.so"))
dev, sr, seconds, nt = 0, 192000, 600.0, 32
s = _dev.stream_ptr(dev)
n = int(sr * seconds); m = int(seconds * sr / 256)
sig = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(L.par_synth_signal_f32(dev, _dev.ptr(sig), 0, n, float(sr), 0x5EED, s))
t = np.linspace(0, seconds, m)
sp = 1.0 + 0.01 * np.sin(2 * np.pi * 0.55 * t + 0.7)
st_t = torch.from_numpy(t * sr).cuda(); sp_t = torch.from_numpy(sp).cuda()
cap = int(n * 1.02) + 1024
nb, ab = int(L.par_speed_plan_bytes(m)), int(L.par_fused_aux_bytes(cap, m))
work = torch.empty(nb, dtype=torch.uint8, device="cuda"); aux = torch.empty(ab, dtype=torch.uint8, device="cuda")
out = torch.empty(cap, dtype=torch.float32, device="cuda")
lo, tr, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
_lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st_t), _dev.ptr(sp_t), m, n, _dev.ptr(work), nb, _dev.ptr(aux), ab, cap,
                                         ctypes.byref(lo), ctypes.byref(tr), 0, None, ctypes.byref(ok), s))
side = torch.cuda.Stream()
def sinc():
    _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(sp_t), m, _dev.ptr(work), _dev.ptr(aux), cap, lo.value, _dev.ptr(sig), 1, n, nt, _dev.ptr(out), 1, s))
def timed(side_fn, reps=30):
    sinc(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        if side_fn:
            side.wait_stream(torch.cuda.current_stream())
            side_fn()
        sinc()
        torch.cuda.current_stream().wait_stream(side)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
sp_ = side.cuda_stream
cases = [("alone", None),
         ("60k empty WGs x 64 thr (1 launch)", lambda: E.launch_empty(60000, 64, ctypes.c_void_p(sp_))),
         ("60k empty WGs x 256 thr", lambda: E.launch_empty(60000, 256, ctypes.c_void_p(sp_))),
         ("20 launches x 3k WGs x 256", lambda: [E.launch_empty(3000, 256, ctypes.c_void_p(sp_)) for _ in range(20)]),
         ("60k WGs x 256 thr, 200 FMAs each", lambda: E.launch_spin(60000, 256, 200, ctypes.c_void_p(sp_))),
         ("6k WGs x 256 thr, 2000 FMAs each", lambda: E.launch_spin(6000, 256, 2000, ctypes.c_void_p(sp_)))]
timed(None, 30)                                      # clocks up
for name, fn in cases + [("alone (again)", None)]:
    print(f"{name:40s} {timed(fn):.3f} ms per K_sinc (115 M outputs)")
