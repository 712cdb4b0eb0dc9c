"""Stereo streaming kernel against the stereo block kernel on interleaved files (NT = 32): per-channel error relative to the
channel's peak, the number of tiles handed back to the block kernel, and both kernels' times."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pyaudiorestoration_amd import _dev, _lib
L = _lib.lib()
dev, sr, seconds = 0, 192000, float(os.environ.get("SECONDS_", "600"))
s = _dev.stream_ptr(dev)
n = int(sr * seconds); m = int(seconds * sr / 256)
mono = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(L.par_synth_signal_f32(dev, _dev.ptr(mono), 0, n, float(sr), 0x5EED, s))
right = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(L.par_synth_signal_f32(dev, _dev.ptr(right), 0, n, float(sr), 0xBEEF, s))
sig = torch.stack((mono, 0.5 * right), dim=1).contiguous().reshape(-1)
t = np.linspace(0, seconds, m)
for name, sp in (("slow 0.990..1.000", 0.995 + 0.005 * np.sin(2 * np.pi * 0.55 * t + 0.7)),
                 ("fast 1.000..1.010", 1.005 + 0.005 * np.sin(2 * np.pi * 0.55 * t + 0.7)),
                 ("bench 0.990..1.010", 1.0 + 0.01 * np.sin(2 * np.pi * 0.55 * t + 0.7))):
    st_t = torch.from_numpy(t * sr).cuda(); sp_t = torch.from_numpy(sp).cuda()
    cap = int(n * 1.02) + 1024
    nb, ab = int(L.par_speed_plan_bytes(m)), int(L.par_fused_aux_bytes(cap, m))
    work = torch.empty(nb, dtype=torch.uint8, device="cuda"); aux = torch.empty(ab, dtype=torch.uint8, device="cuda")
    lo, tr, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st_t), _dev.ptr(sp_t), m, n, _dev.ptr(work), nb, _dev.ptr(aux), ab, cap,
                                             ctypes.byref(lo), ctypes.byref(tr), 0, None, ctypes.byref(ok), s))
    outs, times = [], []
    for form in (0, -1):
        L.par_debug_sinc_kernel(form)
        out = torch.full((cap * 2,), float("nan"), dtype=torch.float32, device="cuda")
        def run():
            _lib.check(L.par_varispeed_fused_stereo_f32(dev, _dev.ptr(sp_t), m, _dev.ptr(work), _dev.ptr(aux), cap, lo.value, _dev.ptr(sig),
                                                        ctypes.c_void_p(sig.data_ptr() + 4), 2, n, 32, _dev.ptr(out), ctypes.c_void_p(out.data_ptr() + 4), 2, s))
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 10)
        outs.append(out[: 2 * lo.value].reshape(-1, 2).clone())
    redo = ctypes.c_int(0)
    _lib.check(L.par_fused_redo_tiles(dev, _dev.ptr(aux), cap, m, ctypes.byref(redo), s))
    a, b = outs
    errs = []
    for c in (0, 1):
        d = (a[:, c] - b[:, c]).abs()
        bad = torch.isnan(d)
        errs.append((float(d[~bad].max() / a[:, c].abs().max()), int(bad.sum()), int(torch.argmax(torch.nan_to_num(d, nan=1e9)))))
    print(f"{name}: block {times[0]:.3f} ms ({2 * lo.value / times[0] / 1e6:.1f} G)  streaming {times[1]:.3f} ms ({2 * lo.value / times[1] / 1e6:.1f} G)  "
          f"redo tiles {redo.value} of {lo.value // 1024}  err L {errs[0][0]:.2e} (nan {errs[0][1]}, at {errs[0][2]})  R {errs[1][0]:.2e} (nan {errs[1][1]}, at {errs[1][2]})", flush=True)
