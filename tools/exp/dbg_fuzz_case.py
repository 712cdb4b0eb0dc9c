import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from oracle import oracle_c as C
from pyaudiorestoration_amd import resampling as R
rng = np.random.default_rng(204836)
n = int(rng.choice([3000, 20000, 150000, 700000])); NT = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 13, 16, 31, 32, 33, 50, 64, 100]))
seg = int(rng.choice([16, 64, 256, 1000, 5000, 60000, 400000])); m = max(2, n // seg); st = np.linspace(0, n, m)
grid = int(rng.integers(0, 4))
if grid == 1: st = st + float(rng.choice([-1000.0, 0.37, 12345.678, 2.0 ** 20 - 3, 2.0 ** 24 - 100]))
elif grid == 2: st = np.concatenate(([0.0], np.cumsum(rng.uniform(0.3, 1.7, m - 1)))) * (n / max(m - 1, 1))
style = int(rng.integers(0, 8))
print(n, NT, seg, m, grid, style, st)
sp = 1.0 + 0.2 * np.sign(np.sin(np.arange(m) * 0.3))
sig = rng.standard_normal(n).astype(np.float32); sig[n // 3:n // 3 + 500] = 0.0; sig[n // 2:] *= np.float32(rng.choice([1.0, 1e-3, 30.0]))
print("speeds", sp)
st_t, sp_t, sig_t = torch.from_numpy(st).cuda(), torch.from_numpy(sp).cuda(), torch.from_numpy(sig).cuda()
ref_pos, _ = C.speed_to_pos(st, sp, n)
plan = R.speed_plan_dev(st_t, sp_t, n, fused=True)
ref = C.sinc(ref_pos, sig, NT, threads=8)
a = R.sinc_resample_dev(torch.from_numpy(ref_pos).cuda(), sig_t, NT).cpu().numpy()
f = R.varispeed_fused_dev(plan, sig_t, NT).cpu().numpy()
sc = np.abs(ref).max()
print("len", len(ref), "pos-array vs oracle", np.abs(a - ref).max() / sc, "fused vs oracle", np.abs(f - ref).max() / sc, "fused vs pos-array", np.abs(f - a).max() / sc, "at", int(np.argmax(np.abs(f - a))), "scale", sc)
i = int(np.argmax(np.abs(f - a))); print("pos there", ref_pos[i], "period", ref_pos[min(i + 1, len(ref_pos) - 1)] - ref_pos[i], "local |ref|", np.abs(ref[max(0, i - 50):i + 50]).max())
for NT2 in (32, 16, 50):
    ref2 = C.sinc(ref_pos, sig, NT2, threads=8); a2 = R.sinc_resample_dev(torch.from_numpy(ref_pos).cuda(), sig_t, NT2).cpu().numpy(); f2 = R.varispeed_fused_dev(plan, sig_t, NT2).cpu().numpy()
    print("NT", NT2, "fused vs pos-array", np.abs(f2 - a2).max() / np.abs(ref2).max(), "fused vs oracle", np.abs(f2 - ref2).max() / np.abs(ref2).max())
