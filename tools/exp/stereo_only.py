"""The stereo K_sinc alone (interleaved 10-min file, NT = 32) on an all-slow, an all-fast and the benchmark's tape."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pyaudiorestoration_amd import _dev, _lib
L = _lib.lib()
if os.environ.get("PAR_BLOCK"):
    L.par_debug_sinc_kernel(0)      # the stereo block kernel instead of two strided streaming launches
dev, sr, seconds, nt = 0, 192000, float(os.environ.get("SECONDS_", "300")), int(os.environ.get("NT", "32"))
s = _dev.stream_ptr(dev)
n = int(sr * seconds); m = int(seconds * sr / 256)
mono = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(L.par_synth_signal_f32(dev, _dev.ptr(mono), 0, n, float(sr), 0x5EED, s))
sig = torch.stack((mono, mono.flip(0)), dim=1).contiguous().reshape(-1)
t = np.linspace(0, seconds, m)
for name, sp in (("slow tape 0.990..1.000 (fc < 1)", 0.995 + 0.005 * np.sin(2 * np.pi * 0.55 * t + 0.7)),
                 ("fast tape 1.000..1.010 (fc = 1)", 1.005 + 0.005 * np.sin(2 * np.pi * 0.55 * t + 0.7)),
                 ("benchmark 0.990..1.010       ", 1.0 + 0.01 * np.sin(2 * np.pi * 0.55 * t + 0.7))):
    st_t = torch.from_numpy(t * sr).cuda(); sp_t = torch.from_numpy(sp).cuda()
    cap = int(n * 1.02) + 1024
    nb, ab = int(L.par_speed_plan_bytes(m)), int(L.par_fused_aux_bytes(cap, m))
    work = torch.empty(nb, dtype=torch.uint8, device="cuda"); aux = torch.empty(ab, dtype=torch.uint8, device="cuda")
    out = torch.empty(cap * 2, dtype=torch.float32, device="cuda")
    lo, tr, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st_t), _dev.ptr(sp_t), m, n, _dev.ptr(work), nb, _dev.ptr(aux), ab, cap,
                                             ctypes.byref(lo), ctypes.byref(tr), 0, None, ctypes.byref(ok), s))
    def run():
        _lib.check(L.par_varispeed_fused_stereo_f32(dev, _dev.ptr(sp_t), m, _dev.ptr(work), _dev.ptr(aux), cap, lo.value, _dev.ptr(sig),
                                                    ctypes.c_void_p(sig.data_ptr() + 4), 2, n, nt, _dev.ptr(out), ctypes.c_void_p(out.data_ptr() + 4), 2, s))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{os.path.basename(os.environ.get('PAR_HIP_LIB','default')):28s} NT={nt} {name}: {ms:.3f} ms  {2 * lo.value / ms / 1e6:.1f} Gsamples/s (both channels)")
