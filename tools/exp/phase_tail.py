"""Per-phase wave clock (PAR_SINC_EXP=64 build) of the FIRST and LAST workgroups of the stereo block kernel on an interleaved
file: what the file's end tiles cost (they are what the tile list behind the streaming kernel waits for)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from pyaudiorestoration_amd import _dev, _lib
L = _lib.lib()
raw = ctypes.CDLL(os.environ["PAR_HIP_LIB"])
L.par_debug_sinc_kernel(0)
dev, sr, seconds = 0, 192000, 300.0
s = _dev.stream_ptr(dev)
n = int(sr * seconds); m = int(seconds * sr / 256)
mono = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(L.par_synth_signal_f32(dev, _dev.ptr(mono), 0, n, float(sr), 0x5EED, s))
sig = torch.stack((mono, mono.flip(0)), dim=1).contiguous().reshape(-1)
st = torch.empty(m, dtype=torch.float64, device="cuda"); sp = torch.empty(m, dtype=torch.float64, device="cuda")
_lib.check(L.par_synth_speed_curve_f64(dev, _dev.ptr(st), _dev.ptr(sp), m, seconds, float(sr), 0.01, 0.55, 0.7, s))
cap = int(n * 1.02) + 1024
nb, ab = int(L.par_speed_plan_bytes(m)), int(L.par_fused_aux_bytes(cap, m))
work = torch.empty(nb, dtype=torch.uint8, device="cuda"); aux = torch.empty(ab, dtype=torch.uint8, device="cuda")
out = torch.empty(cap * 2, dtype=torch.float32, device="cuda")
lo, tr, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
_lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st), _dev.ptr(sp), m, n, _dev.ptr(work), nb, _dev.ptr(aux), ab, cap,
                                         ctypes.byref(lo), ctypes.byref(tr), 0, None, ctypes.byref(ok), s))
groups = (lo.value + 511) // 512                     # stereo: 4 waves x 128 outputs per workgroup
dbg = torch.zeros((groups * 4, 8), dtype=torch.int32, device="cuda")
assert raw.par_debug_sinc_phase_buffer(ctypes.c_void_p(dbg.data_ptr())) == 0
_lib.check(L.par_varispeed_fused_stereo_f32(dev, _dev.ptr(sp), m, _dev.ptr(work), _dev.ptr(aux), cap, lo.value, _dev.ptr(sig),
                                            ctypes.c_void_p(sig.data_ptr() + 4), 2, n, 32, _dev.ptr(out), ctypes.c_void_p(out.data_ptr() + 4), 2, s))
torch.cuda.synchronize()
rows = dbg.cpu().numpy()
print(f"len_out {lo.value} (plan kind {ok.value}), {groups} workgroups; last partial wave holds {lo.value % 128} outputs; cycles at 100 MHz (s_memtime)")
print("phases: issue | placement | span->LDS | barrier | taps | stores+slow path")
for g in list(range(0, 2)) + list(range(groups - 4, groups)):
    for w in range(4):
        print(f"  group {g:7d} wave {w}: " + " ".join(f"{int(v):8d}" for v in rows[g * 4 + w][:6]) + f"   flags {int(rows[g * 4 + w][6]) & 255:04b} slow outputs {int(rows[g * 4 + w][6]) >> 8} span {int(rows[g * 4 + w][7])}")
med = np.median(rows[4 * 1000: 4 * 2000, :6], axis=0)
print("  median of groups 1000-2000:   " + " ".join(f"{int(v):8d}" for v in med))
