"""resampling.run's one download per file: a fresh pinned buffer per file (r02-r05) against the staged ring (_dev.to_host, r06)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pyaudiorestoration_amd import _dev
for n in (115_200_000, 2 * 115_200_000):
    out_t = torch.randn(n, device="cuda")
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        host = torch.empty(out_t.shape, dtype=torch.float32, pin_memory=True); host.copy_(out_t); a = host.numpy()
        t1 = time.perf_counter()
        b = _dev.to_host(out_t)
        t2 = time.perf_counter()
        print(f"{n} floats: fresh pinned buffer {1e3 * (t1 - t0):.1f} ms, staged ring {1e3 * (t2 - t1):.1f} ms", flush=True)
        del host, a, b
