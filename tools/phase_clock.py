"""Per-phase wave clock of the fused K_sinc (experiment build with PAR_SINC_EXP bit 64):
    python tools/build_variant.py tools/ab/libpar_clk.so -DPAR_SINC_EXP=64
    PAR_HIP_LIB=$PWD/tools/ab/libpar_clk.so python tools/phase_clock.py
Prints the average cycles a wave spends in each phase of one tile (s_memtime differences, one lane per wave)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pyaudiorestoration_amd import _dev, _lib

L = _lib.lib()
raw = ctypes.CDLL(os.environ["PAR_HIP_LIB"])
dev, sr, seconds, nt = 0, 192000, float(sys.argv[1]) if len(sys.argv) > 1 else 600.0, 32
s = _dev.stream_ptr(dev)
n = int(sr * seconds)
m = int(seconds * sr / 256)
sig = torch.empty(n, dtype=torch.float32, device="cuda")
st = torch.empty(m, dtype=torch.float64, device="cuda")
sp = torch.empty(m, dtype=torch.float64, device="cuda")
_lib.check(L.par_synth_signal_f32(dev, _dev.ptr(sig), 0, n, float(sr), 0x5EED, s))
_lib.check(L.par_synth_speed_curve_f64(dev, _dev.ptr(st), _dev.ptr(sp), m, seconds, float(sr), 0.01, 0.55, 0.7, s))
cap = int(n * 1.02) + 1024
nb, ab = int(L.par_speed_plan_bytes(m)), int(L.par_fused_aux_bytes(cap, m))
work = torch.empty(nb, dtype=torch.uint8, device="cuda")
aux = torch.empty(ab, dtype=torch.uint8, device="cuda")
out = torch.empty(cap, dtype=torch.float32, device="cuda")
lo, tr, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
_lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st), _dev.ptr(sp), m, n, _dev.ptr(work), nb, _dev.ptr(aux), ab, cap,
                                         ctypes.byref(lo), ctypes.byref(tr), 0, None, ctypes.byref(ok), s))
waves = ((lo.value + 1023) // 1024) * 4
dbg = torch.zeros((waves, 8), dtype=torch.int32, device="cuda")
assert raw.par_debug_sinc_phase_buffer(ctypes.c_void_p(dbg.data_ptr())) == 0
reps = 3
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(sp), m, _dev.ptr(work), _dev.ptr(aux), cap, lo.value, _dev.ptr(sig), 1, n,
                                         nt, _dev.ptr(out), 1, s))
e1.record()
torch.cuda.synchronize()
buf = dbg.to(torch.float64).sum(dim=0).cpu().numpy()          # the last launch's rows
names = ["issue loads", "placement (+wait records/header)", "span->LDS (+wait signal)", "barrier", "taps", "stores"]
tot = sum(buf[k] for k in range(6))
print(f"{lo.value} outputs, {e0.elapsed_time(e1) / reps:.3f} ms per launch, {waves} waves")
for k, nm in enumerate(names):
    print(f"  {nm:36s} {buf[k] / waves:9.1f} cycles/wave  {100.0 * buf[k] / tot:5.1f} %")
print(f"  {'wave lifetime (sum)':36s} {tot / waves:9.1f} cycles")
