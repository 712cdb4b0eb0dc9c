#!/usr/bin/env python3
"""Config-3 pilot flow on the GPU against tests/golden/pipeline.npz: where the end-to-end difference comes from."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import inputs
from pyaudiorestoration_amd import pipeline

g = np.load(os.path.join(ROOT, "tests", "golden", "pipeline.npz"))
sr, n, n_fft, hop = (int(v) for v in g["cfg"])
x = inputs.pilot(n, sr)
r = pipeline.respeed(x, sr, [(0.05, 4000.0), (1.45, 4000.0)], n_fft, hop, 1, "Peak", 0.5, (0, 20), 32)
f, c = r["freqs"], r["speed_curve"]
pos, y = r["positions"].cpu().numpy(), r["output"].cpu().numpy()[:, 0]
e = np.abs(y - g["y"]) / np.max(np.abs(g["y"]))
print(f"freqs rel {np.max(np.abs(f - g['track_freqs']) / g['track_freqs']):.2e}  curve rel "
      f"{np.max(np.abs(c[:, 1] - g['curve'][:, 1]) / g['curve'][:, 1]):.2e}  pos abs {np.max(np.abs(pos - g['pos'])):.2e}  "
      f"output rel max {e.max():.2e} median {np.median(e):.2e} samples over 1e-5: {(e > 1e-5).sum()} of {len(e)}")
d = pos - g["pos"]
print("pos diff quantiles", np.quantile(d, [0, 0.25, 0.5, 0.75, 1]))
