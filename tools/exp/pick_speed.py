"""One channel of a two-channel interleaved file (k_sinc_pipe<2, 3>) against the block kernel and against the stereo launch: ms per launch."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pyaudiorestoration_amd import _dev, _lib, resampling
L = _lib.lib()
sr, seconds = 192000, float(sys.argv[1]) if len(sys.argv) > 1 else 3600.0
n, m = int(sr * seconds), int(seconds * sr / 256)
s = _dev.stream_ptr(0)
mono = torch.empty(n, dtype=torch.float32, device="cuda")
sig = torch.empty((n, 2), dtype=torch.float32, device="cuda")
for c in range(2):
    _lib.check(L.par_synth_signal_f32(0, _dev.ptr(mono), 0, n, float(sr), 0x5EED + c, s))
    sig[:, c] = mono
st = torch.empty(m, dtype=torch.float64, device="cuda"); sp = torch.empty(m, dtype=torch.float64, device="cuda")
_lib.check(L.par_synth_speed_curve_f64(0, _dev.ptr(st), _dev.ptr(sp), m, seconds, float(sr), 0.01, 0.55, 0.7, s))
plan = resampling.speed_plan_dev(st, sp, n, fused=True)
flat = sig.reshape(-1)
out = torch.empty(plan.len_out, dtype=torch.float32, device="cuda")
def timed(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for form, name in ((-1, "streaming (default)"), (0, "block kernel")):
    L.par_debug_sinc_kernel(form)
    ms1 = timed(lambda: resampling.varispeed_fused_dev(plan, flat[1:], 32, out, sig_stride=2, len_in=n))
    msm = timed(lambda: resampling.varispeed_fused_dev(plan, mono, 32, out))
    print(f"{name:22s} one channel of two: {ms1:.3f} ms = {plan.len_out / ms1 / 1e6:.1f} G channel-samples/s; mono unit stride: {msm:.3f} ms = {plan.len_out / msm / 1e6:.1f} G")
L.par_debug_sinc_kernel(-1)
