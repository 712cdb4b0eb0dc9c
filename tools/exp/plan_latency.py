"""How long the plan takes (host wall time of par_speed_to_pos_plan_fused on a side stream, which ends with a header read-back)
while K_sinc of the 60-min file runs on the main stream: the step of the pipelined bench is max(K_sinc, this)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pyaudiorestoration_amd import _dev, _lib
L = _lib.lib()
dev, sr, seconds, nt = 0, 192000, 3600.0, 32
s = _dev.stream_ptr(dev)
side = torch.cuda.Stream(device=dev, priority=int(os.environ.get("PAR_SIDE_PRIO", "0")))
sp_side = ctypes.c_void_p(side.cuda_stream)
n = int(sr * seconds); m = int(seconds * sr / 256)
sig = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(L.par_synth_signal_f32(dev, _dev.ptr(sig), 0, n, float(sr), 0x5EED, s))
st = torch.empty(m, dtype=torch.float64, device="cuda"); sp = torch.empty(m, dtype=torch.float64, device="cuda")
_lib.check(L.par_synth_speed_curve_f64(dev, _dev.ptr(st), _dev.ptr(sp), m, seconds, float(sr), 0.01, 0.55, 0.7, s))
cap = int(n * 1.02) + 1024
nb, ab = int(L.par_speed_plan_bytes(m)), int(L.par_fused_aux_bytes(cap, m))
work = [torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in range(2)]
aux = [torch.empty(ab, dtype=torch.uint8, device="cuda") for _ in range(2)]
out = torch.empty(cap, dtype=torch.float32, device="cuda")
lo, tr, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
def plan(slot, stream):
    _lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st), _dev.ptr(sp), m, n, _dev.ptr(work[slot]), nb, _dev.ptr(aux[slot]), ab, cap,
                                             ctypes.byref(lo), ctypes.byref(tr), 0, None, ctypes.byref(ok), stream))
plan(0, s)
len_out = lo.value
def sinc():
    _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(sp), m, _dev.ptr(work[0]), _dev.ptr(aux[0]), cap, len_out, _dev.ptr(sig), 1, n, nt, _dev.ptr(out), 1, s))
for _ in range(3): sinc()
torch.cuda.synchronize()
t = []
for _ in range(5):
    t0 = time.perf_counter(); plan(1, sp_side); t.append(time.perf_counter() - t0)
print(f"plan alone: {np.median(t) * 1e3:.2f} ms")
t = []
for _ in range(8):
    sinc(); sinc()                       # keep the GPU busy for ~10 ms
    t0 = time.perf_counter(); plan(1, sp_side); t.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print(f"plan under K_sinc: {np.median(t) * 1e3:.2f} ms  (all: {' '.join(f'{x * 1e3:.2f}' for x in t)})")
