"""How much host CPU does one rank of the archive flow burn?  (process CPU time / wall time over 96 stereo 10-min files)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pyaudiorestoration_amd import _dev, _lib, resampling
L = _lib.lib()
sr, seconds = 192000, 600.0
n, m = int(sr * seconds), int(seconds * sr / 256)
s = _dev.stream_ptr(0)
sig = torch.empty((n, 2), dtype=torch.float32, device="cuda")
_lib.check(L.par_synth_signal_f32(0, _dev.ptr(sig), 0, 2 * n, float(sr), 0x5EED, s))
st = torch.empty(m, dtype=torch.float64, device="cuda"); sp = torch.empty(m, dtype=torch.float64, device="cuda")
_lib.check(L.par_synth_speed_curve_f64(0, _dev.ptr(st), _dev.ptr(sp), m, seconds, float(sr), 0.01, 0.55, 0.7, s))
items = [(st, sp, sig)] * 96
for rep in range(3):
    torch.cuda.synchronize()
    w0, c0 = time.perf_counter(), time.process_time()
    for _ in resampling.varispeed_batch_dev(items, 32, 0): pass
    torch.cuda.synchronize()
    w, c = time.perf_counter() - w0, time.process_time() - c0
    print(f"wall {w * 1e3:.1f} ms, process CPU {c * 1e3:.1f} ms = {c / w:.2f} cores busy; {w * 1e3 / 96:.3f} ms per file", flush=True)
