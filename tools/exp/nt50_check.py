"""NT = 50 through the fused entry point against the C oracle: fast (fc = 1: the Farrow bank on the matrix cores since r06), slow, mixed
tapes x noise / tones / a quiet passage / samples float16 does not suit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import oracle_c as C
from pyaudiorestoration_amd import resampling
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200000
m = n // 256
st = np.linspace(0, n, m); t = st / 192000.0
rng = np.random.default_rng(5); tt = np.arange(n)
signals = {"white noise": rng.standard_normal(n).astype(np.float32), "Nyquist tone": np.cos(np.pi * tt).astype(np.float32),
           "0.45 fs": np.cos(0.9 * np.pi * tt + 0.2).astype(np.float32),
           "quiet third": (rng.standard_normal(n) * np.where((tt > n // 3) & (tt < 2 * n // 3), 1e-3, 1.0)).astype(np.float32),
           "x 1e4": (1e4 * rng.standard_normal(n)).astype(np.float32)}
worst = 0.0
for cname, sp in (("fast 1.000..1.010", 1.005 + 0.005 * np.sin(2 * np.pi * 4.4 * t + 0.7)), ("slow", 0.995 + 0.00499 * np.sin(2 * np.pi * 4.4 * t + 0.7)),
                  ("mix", 1.0 + 0.01 * np.sin(2 * np.pi * 4.4 * t + 0.7)), ("constant 1.02", np.full(m, 1.02))):
    pos, _ = C.speed_to_pos(st, sp, n)
    plan = resampling.speed_plan_dev(torch.from_numpy(st).cuda(), torch.from_numpy(sp).cuda(), n, fused=True)
    for NT in (50, 32):
        for name, sig in signals.items():
            want = C.sinc(pos, sig, NT, threads=16)
            got = resampling.varispeed_fused_dev(plan, torch.from_numpy(sig).cuda(), NT).cpu().numpy()
            pk = np.max(np.abs(want)); e = float(np.nanmax(np.abs(got - want)) / pk); bad = int(np.isnan(got).sum())
            k = (len(want) // 4096) * 4096
            bw, be = np.abs(want[:k]).reshape(-1, 4096).max(1), np.abs(got[:k] - want[:k]).reshape(-1, 4096).max(1)
            blk = float(np.max(np.where(bw > 1e-4 * pk, be / np.maximum(bw, 1e-30), 0.0)))
            worst = max(worst, e if not bad else 1.0)
            print(f"{cname:18s} NT {NT} {name:14s} max|err|/peak {e:.2e}  block-relative {blk:.2e}  nan {bad}", flush=True)
print("worst", worst)
