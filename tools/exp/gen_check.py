"""fc < 1 tiles through k_sinc_fused<1, 32, 4, GEN> against the C oracle: slow tape (speed 0.990 .. 0.99999), 2 M samples."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import oracle_c as C
from pyaudiorestoration_amd import resampling as R
n = 2000000
m = n // 256
st = np.linspace(0, n, m)
t = st / 192000.0
rng = np.random.default_rng(5)
tt = np.arange(n)
signals = {"white noise": rng.standard_normal(n).astype(np.float32),
           "full-scale Nyquist tone": np.cos(np.pi * tt).astype(np.float32),
           "full-scale tone at 0.45 fs": np.cos(0.9 * np.pi * tt + 0.2).astype(np.float32)}
for cname, sp in (("0.990..1.000 wow", 0.995 + 0.00499 * np.sin(2 * np.pi * 0.55 * t * 8 + 0.7)),
                  ("0.97..0.99 fast wow", 0.98 + 0.01 * np.sin(2 * np.pi * 4.0 * t + 0.3)),
                  ("benchmark mix", 1.0 + 0.01 * np.sin(2 * np.pi * 0.55 * t * 8 + 0.7))):
    pos, _ = C.speed_to_pos(st, sp, n)
    plan = R.speed_plan_dev(torch.from_numpy(st).cuda(), torch.from_numpy(sp).cuda(), n, fused=True)
    for name, sig in signals.items():
        want = C.sinc(pos, sig, 32, threads=32)
        got = R.varispeed_fused_dev(plan, torch.from_numpy(sig).cuda(), 32).cpu().numpy()
        pk = np.max(np.abs(want))
        err = np.abs(got - want) / pk
        print(f"{cname:22s} {name:28s} max |err| / peak = {err.max():.2e} at {int(err.argmax())}  (len {len(want)}, nan {int(np.isnan(got).sum())})")
