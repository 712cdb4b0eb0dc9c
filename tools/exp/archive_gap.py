"""Why does the 48-file archive measurement of the default bench line (same item 48 times) run below the config-5 flow?
Variants of the item source, one process: python tools/exp/archive_gap.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pyaudiorestoration_amd import _dev, _lib, resampling
L = _lib.lib()
sr, seconds = 192000, 600.0
n, m = int(sr * seconds), int(seconds * sr / 256)
s = _dev.stream_ptr(0)
mono = torch.empty(n, dtype=torch.float32, device="cuda")
ring = []
for k in range(6):
    sig = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    for c in range(2):
        _lib.check(L.par_synth_signal_f32(0, _dev.ptr(mono), 0, n, float(sr), 0x5EED + 2 * k + c, s))
        sig[:, c] = mono
    ring.append(sig)
curves = torch.empty((32, 2, m), dtype=torch.float64, device="cuda")
def synth(k, phase):
    c = curves[k % 32]
    _lib.check(L.par_synth_speed_curve_f64(0, _dev.ptr(c[0]), _dev.ptr(c[1]), m, seconds, float(sr), 0.01, 0.55, phase, s))
    return c[0], c[1]
st0, sp0 = synth(0, 0.7)
st0, sp0 = st0.clone(), sp0.clone()
F = 48
def v_list():          return [(st0, sp0, ring[0])] * F
def v_gen_same():
    for k in range(F): yield st0, sp0, ring[0]
def v_gen_ring():
    for k in range(F): yield st0, sp0, ring[k % 6]
def v_gen_curves_same_phase():
    for k in range(F):
        a, b = synth(k, 0.7); yield a, b, ring[k % 6]
def v_gen_curves():
    for k in range(F):
        a, b = synth(k, 0.7 + k); yield a, b, ring[k % 6]
def v_gen_curves_one_sig():
    for k in range(F):
        a, b = synth(k, 0.7 + k); yield a, b, ring[0]
for name, v in (("list, same item", v_list), ("generator, same item", v_gen_same), ("ring of 6 signals, same curve tensors", v_gen_ring),
                ("ring + curve made per file, same phase", v_gen_curves_same_phase), ("ring + curve per file, phase 0.7 + k (config 5)", v_gen_curves),
                ("one signal + curve per file, phase 0.7 + k", v_gen_curves_one_sig), ("list, same item", v_list)):
    best, reps = 1e9, []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tot = 0
        for _, out, plan in resampling.varispeed_batch_dev(v(), 32, 0): tot += 2 * plan.len_out
        torch.cuda.synchronize(); reps.append(time.perf_counter() - t0); best = min(best, reps[-1])
    print(f"{name:50s} {best * 1e3 / F:.3f} ms per file = {tot / best / 1e9:.1f} G   reps: " + " ".join(f"{r * 1e3 / F:.3f}" for r in reps))
import bench
for rep in range(2):
    r = bench.stereo_secondary(0)
    print(f"bench.stereo_secondary: serial {r['ms_per_file']} ms, batched {r['batched_ms_per_file']} ms = {r['batched_Msamples/s']} M/s", flush=True)
best, reps = 1e9, []
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _, out, plan in resampling.varispeed_batch_dev(v_list(), 32, 0): pass
    torch.cuda.synchronize(); reps.append(time.perf_counter() - t0)
print("list, same item, after stereo_secondary: " + " ".join(f"{r * 1e3 / F:.3f}" for r in reps))
