"""Grouped (merged multi-file) K_sinc launches against file-by-file launches: (a) K_sinc alone on pre-made plans, (b) the batch
driver (plans included).  python tools/exp/group_speed.py SECONDS [stereo]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pyaudiorestoration_amd import _dev, _lib, resampling
L = _lib.lib()
sr, seconds = 192000, float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
stereo = len(sys.argv) > 2 and sys.argv[2] == "stereo"
n, m = int(sr * seconds), max(int(seconds * sr / 256), 16)
s = _dev.stream_ptr(0)
F = 8
mono = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(L.par_synth_signal_f32(0, _dev.ptr(mono), 0, n, float(sr), 0x5EED, s))
sig = torch.stack([mono, mono], 1).contiguous() if stereo else mono
items, plans = [], []
for k in range(F):
    st = torch.empty(m, dtype=torch.float64, device="cuda"); sp = torch.empty(m, dtype=torch.float64, device="cuda")
    _lib.check(L.par_synth_speed_curve_f64(0, _dev.ptr(st), _dev.ptr(sp), m, seconds, float(sr), 0.01, 0.55, 0.7 + k, s))
    items.append((st, sp, sig))
    plans.append(resampling.speed_plan_dev(st, sp, n, fused=True))
ch = 2 if stereo else 1
tot = sum(p.len_out for p in plans) * ch
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
def single():
    for p, it in zip(plans, items): resampling._resample_item(p, it, 32, 0)
print(f"{seconds:.0f} s {'stereo' if stereo else 'mono'} x {F} files, K_sinc alone on ready plans:")
ms = timed(single); print(f"  file by file : {ms:.3f} ms = {tot / ms / 1e6:.1f} G")
for g in (2, 4, 8):
    def grouped():
        for a in range(0, F, g): resampling._resample_group(plans[a:a + g], items[a:a + g], 32, 0)
    ms = timed(grouped); print(f"  groups of {g}  : {ms:.3f} ms = {tot / ms / 1e6:.1f} G")
print(f"batch driver (plans included), 64 files, PAR_PLANNERS={os.environ.get('PAR_PLANNERS', '3')}:")
def gen():
    for k in range(64): yield items[k % F]
for g in (1, 2, 4, 8):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in resampling.varispeed_batch_dev(gen(), 32, 0, group=g): pass
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"  group {g}: {dt * 1e3 / 64:.3f} ms per file = {tot / F * 64 / dt / 1e9:.1f} G")
