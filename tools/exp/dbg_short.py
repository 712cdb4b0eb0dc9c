import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch as t, inputs
from oracle import oracle_c as C
from pyaudiorestoration_amd import resampling as R
rng = np.random.default_rng(3)
st = np.linspace(0, 600000, 600000 // 64); rng.standard_normal(len(st))
st = np.cumsum(rng.uniform(3.0, 40.0, 30000)); sp = rng.uniform(0.8, 1.25, 30000)
n_sig, n_in = int(st[-1]) + 100, int(st[-1] * 0.7)
sig = inputs.noise(n_sig, 5)
pos, _ = C.speed_to_pos(st, sp, n_in)
st_t, sp_t, sig_t = t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), t.from_numpy(sig).cuda()
plan = R.speed_plan_dev(st_t, sp_t, n_in, fused=True)
out = R.varispeed_fused_dev(plan, sig_t, 32).cpu().numpy()
pos_t = R.speed_to_pos_dev(st_t, sp_t, n_in)
out2 = R.sinc_resample_dev(pos_t, sig_t, 32).cpu().numpy()
ref = C.sinc(pos, sig, 32, threads=8)
print(len(pos), plan.len_out, np.isnan(out).sum(), np.isnan(out2).sum(), np.isnan(ref).sum(), np.array_equal(pos_t.cpu().numpy(), pos))
i = 660809
print(np.abs(out[i:] - ref[i:]).max(), np.abs(out2[i:] - ref[i:]).max(), np.abs(ref[i:]).max(), pos[i], pos[-1], n_sig)
w = C.sinc(pos[i:i + 1201], sig, 32)
print(np.isnan(w).sum(), np.abs(w - ref[i:i + 1201]).max())
