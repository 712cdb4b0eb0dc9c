"""Streaming K_sinc (sinc2.hip) against the C oracle: slow / fast / benchmark-mix tapes x noise / Nyquist / 0.45 fs, 2 M samples;
prints the worst error and how many tiles went back to the block kernel.  par_debug_sinc_kernel(0): the block kernel alone."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import oracle_c as C
from pyaudiorestoration_amd import _dev, _lib
L = _lib.lib()
dev = 0
s = _dev.stream_ptr(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
m = n // 256
st = np.linspace(0, n, m)
t = st / 192000.0
rng = np.random.default_rng(5)
tt = np.arange(n)
signals = {"white noise": rng.standard_normal(n).astype(np.float32),
           "full-scale Nyquist tone": np.cos(np.pi * tt).astype(np.float32),
           "tone at 0.45 fs": np.cos(0.9 * np.pi * tt + 0.2).astype(np.float32),
           "quiet third": (rng.standard_normal(n) * np.where((tt > n // 3) & (tt < 2 * n // 3), 1e-3, 1.0)).astype(np.float32)}
worst = 0.0
for cname, sp in (("fast 1.000..1.010", 1.005 + 0.005 * np.sin(2 * np.pi * 0.55 * t * 8 + 0.7)),
                  ("slow 0.990..1.000", 0.995 + 0.00499 * np.sin(2 * np.pi * 0.55 * t * 8 + 0.7)),
                  ("0.97..0.99 fast wow", 0.98 + 0.01 * np.sin(2 * np.pi * 4.0 * t + 0.3)),
                  ("benchmark mix", 1.0 + 0.01 * np.sin(2 * np.pi * 0.55 * t * 8 + 0.7)),
                  ("constant 1.0", np.ones(m)), ("constant 0.999", np.full(m, 0.999))):
    pos, _ = C.speed_to_pos(st, sp, n)
    st_t = torch.from_numpy(st).cuda(); sp_t = torch.from_numpy(sp).cuda()
    cap = int(n * 1.04) + 2048
    nb, ab = int(L.par_speed_plan_bytes(m)), int(L.par_fused_aux_bytes(cap, m))
    work = torch.empty(nb, dtype=torch.uint8, device="cuda"); aux = torch.empty(ab, dtype=torch.uint8, device="cuda")
    out = torch.zeros(cap, dtype=torch.float32, device="cuda")
    lo, tr, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st_t), _dev.ptr(sp_t), m, n, _dev.ptr(work), nb, _dev.ptr(aux), ab, cap,
                                             ctypes.byref(lo), ctypes.byref(tr), 0, None, ctypes.byref(ok), s))
    assert ok.value in (1, 2) and lo.value == len(pos), (ok.value, lo.value, len(pos))
    for name, sig in signals.items():
        want = C.sinc(pos, sig, 32, threads=32)
        sg = torch.from_numpy(sig).cuda()
        out.fill_(float("nan"))
        _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(sp_t), m, _dev.ptr(work), _dev.ptr(aux), cap, lo.value, _dev.ptr(sg), 1, n, 32,
                                             _dev.ptr(out), 1, s))
        redo = ctypes.c_int(0)
        _lib.check(L.par_fused_redo_tiles(dev, _dev.ptr(aux), cap, m, ctypes.byref(redo), s))
        got = out[:lo.value].cpu().numpy()
        pk = np.max(np.abs(want))
        err = np.abs(got - want) / pk
        bad = int(np.isnan(got).sum())
        nb_ = 4096
        k = (len(want) // nb_) * nb_
        bw, be = np.abs(want[:k]).reshape(-1, nb_).max(1), np.abs(got[:k] - want[:k]).reshape(-1, nb_).max(1)
        blk = np.max(np.where(bw > 1e-4 * pk, be / np.maximum(bw, 1e-30), 0.0))
        worst = max(worst, float(np.nanmax(err)) if bad == 0 else 1.0)
        print(f"{cname:20s} {name:24s} max|err|/peak {np.nanmax(err):.2e} at {int(np.nanargmax(err))}  block-relative {blk:.2e}  nan {bad}  "
              f"redo tiles {redo.value}/{(lo.value + 1023) // 1024}", flush=True)
print("worst", worst)
