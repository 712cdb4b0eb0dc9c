#!/bin/bash
# order-3 band-pass across lengths for each library variant given (one gpurun session)
cat > /tmp/_ab_filt.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from pyaudiorestoration_amd import filters
out = []
for n in (1_000_000, 4_000_000, 10_000_000, 16_000_000, 24_000_000, 30_000_000, 36_000_000, 40_000_000, 60_000_000, 80_000_000, 100_000_000):
    x = torch.randn(n, dtype=torch.float64, device="cuda")
    filters.bandpass_dev(x, 300.0, 6000.0, 48000.0, 3)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        filters.bandpass_dev(x, 300.0, 6000.0, 48000.0, 3)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    out.append(f"{n // 1000000}M {best * 1e3:.2f}")
    del x
print("  ".join(out))
PY
for L in "$@"; do
  echo "== $L"
  PAR_HIP_LIB=$GRAFT_REPO_ROOT/$L python /tmp/_ab_filt.py 2>&1 | grep -v amdgpu
done
