"""k_sinc_pipe timing build (-DPAR_S2_EXP=64): per wave, cycles per loop iteration spent in the memory wait at its head, in the body,
and outside the loop."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pyaudiorestoration_amd import _dev, _lib
L = _lib.lib()
dev, sr, seconds, nt = 0, 192000, 600.0, 32
s = _dev.stream_ptr(dev)
n = int(sr * seconds); m = int(seconds * sr / 256)
sig = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(L.par_synth_signal_f32(dev, _dev.ptr(sig), 0, n, float(sr), 0x5EED, s))
t = np.linspace(0, seconds, m)
for name, sp in (("slow", 0.995 + 0.005 * np.sin(2 * np.pi * 0.55 * t + 0.7)), ("fast", 1.005 + 0.005 * np.sin(2 * np.pi * 0.55 * t + 0.7))):
    st_t = torch.from_numpy(t * sr).cuda(); sp_t = torch.from_numpy(sp).cuda()
    cap = int(n * 1.02) + 1024
    nb, ab = int(L.par_speed_plan_bytes(m)), int(L.par_fused_aux_bytes(cap, m))
    work = torch.empty(nb, dtype=torch.uint8, device="cuda"); aux = torch.empty(ab, dtype=torch.uint8, device="cuda")
    out = torch.empty(cap, dtype=torch.float32, device="cuda")
    lo, tr, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st_t), _dev.ptr(sp_t), m, n, _dev.ptr(work), nb, _dev.ptr(aux), ab, cap,
                                             ctypes.byref(lo), ctypes.byref(tr), 0, None, ctypes.byref(ok), s))
    waves = (lo.value // 1024 + 7) // 8 + 1
    buf = torch.zeros(waves * 16, dtype=torch.int64, device="cuda")
    L.par_debug_s2_phase_buffer.argtypes = [ctypes.c_void_p]
    assert L.par_debug_s2_phase_buffer(_dev.ptr(buf)) == 0
    for _ in range(2):
        _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(sp_t), m, _dev.ptr(work), _dev.ptr(aux), cap, lo.value, _dev.ptr(sig), 1, n, nt, _dev.ptr(out), 1, s))
    torch.cuda.synchronize()
    b = buf.cpu().numpy().reshape(-1, 16).astype(np.float64)
    b = b[b[:, 2] > 0]
    it = b[:, 2].sum()
    print(f"{name} tape: {len(b)} waves with loop iterations, {it / len(b):.1f} per wave; per iteration: memory wait {b[:, 0].sum() / it:.0f} cycles, "
          f"body {b[:, 1].sum() / it:.0f}, whole wave life / iterations {b[:, 3].sum() / it:.0f}")
