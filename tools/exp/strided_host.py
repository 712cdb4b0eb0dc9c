"""Column views of interleaved host files (what the reference hands over, util/resampling.py:222-227): one core against the staging
threads for the gather in front of an upload and the scatter behind a download."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from pyaudiorestoration_amd import _dev
n = 115_200_000
a = np.random.default_rng(0).standard_normal((n, 2)).astype(np.float32)
v = a[:, 1]
out = np.zeros((n, 2), np.float32)
for rep in range(3):
    t0 = time.perf_counter(); c = np.ascontiguousarray(v); t1 = time.perf_counter()
    c2 = _dev.contiguous(v); t2 = time.perf_counter()
    out[:, 0] = c; t3 = time.perf_counter()
    _dev.host_assign(out[:, 1], c); t4 = time.perf_counter()
    print(f"gather: one core {t1 - t0:.3f} s, threads {t2 - t1:.3f} s;  scatter: one core {t3 - t2:.3f} s, threads {t4 - t3:.3f} s", flush=True)
