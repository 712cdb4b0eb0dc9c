"""K_sinc at NT = 50 (the reference's default quality) and NT = 32, by tape: fast (fc = 1 everywhere), slow (fc < 1), the benchmark's mix."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pyaudiorestoration_amd import _dev, _lib, resampling
L = _lib.lib()
sr, seconds = 192000, float(sys.argv[1]) if len(sys.argv) > 1 else 1800.0
n, m = int(sr * seconds), int(seconds * sr / 256)
s = _dev.stream_ptr(0)
sig = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(L.par_synth_signal_f32(0, _dev.ptr(sig), 0, n, float(sr), 0x5EED, s))
tt = torch.linspace(0, seconds, m, dtype=torch.float64, device="cuda")
st = tt * sr
def timed(fn, reps=6):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for name, sp in (("fast 1.000..1.010", 1.005 + 0.005 * torch.sin(2 * np.pi * 0.55 * tt + 0.7)), ("slow 0.990..1.000", 0.995 + 0.00499 * torch.sin(2 * np.pi * 0.55 * tt + 0.7)),
                 ("benchmark mix", 1.0 + 0.01 * torch.sin(2 * np.pi * 0.55 * tt + 0.7))):
    plan = resampling.speed_plan_dev(st.contiguous(), sp.contiguous(), n, fused=True)
    out = torch.empty(plan.len_out, dtype=torch.float32, device="cuda")
    row = []
    for NT in (32, 50, 100):
        ms = timed(lambda: resampling.varispeed_fused_dev(plan, sig, NT, out))
        row.append(f"NT {NT}: {ms:.3f} ms = {plan.len_out / ms / 1e6:.1f} G")
    print(f"{name:20s} " + "   ".join(row))
