"""K_sinc back to back for ~10 s (10-min mono file, benchmark curve): per-second throughput and what rocm-smi says about clocks / power."""
import ctypes, os, sys, subprocess, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pyaudiorestoration_amd import _dev, _lib
L = _lib.lib()
dev, sr, seconds, nt = 0, 192000, float(sys.argv[1]) if len(sys.argv) > 1 else 600.0, 32
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
def sinc():
    _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(sp_t), m, _dev.ptr(work), _dev.ptr(aux), cap, lo.value, _dev.ptr(sig), 1, n, nt, _dev.ptr(out), 1, s))
sinc(); torch.cuda.synchronize()
for sec in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(int(600000 / seconds)): sinc()
    e1.record()
    time.sleep(0.4)                      # the queue holds ~0.8 s of kernels: sample the sensors while they run
    smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
    torch.cuda.synchronize()
    g = lambda pat: (re.search(pat, smi) or [None, "?"])[1]
    sclk, pw, tj = g(r"sclk clock level: \S+ \((\d+)Mhz"), g(r"Power \(W\): ([\d.]+)"), g(r"junction\) \(C\): ([\d.]+)")
    print(f"second {sec}: {e0.elapsed_time(e1) / int(600000 / seconds):.4f} ms per launch  sclk {sclk} MHz  power {pw} W  junction {tj} C", flush=True)
