#!/usr/bin/env python3
"""Randomised check of the merged multi-file launches (par_varispeed_fused_batch_f32, r06) on the GPU box: random batches of mono /
interleaved-stereo files (sizes 3 000 .. 3 M samples, gentle / flutter / fast / slow / step curves, silence and full-scale steps),
random group sizes -- every output must be BIT-identical to the file-by-file launches of the same batch driver, and one file per
batch is checked against the C oracle.   python tools/fuzz_groups.py SECONDS [SEED]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import oracle_c as C
from pyaudiorestoration_amd import resampling as R

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t_end = time.time() + budget
batches = files = merged = 0
worst = 0.0
while time.time() < t_end:
    rng = np.random.default_rng(seed0 + batches)
    nf = int(rng.integers(2, 13))
    items, host = [], []
    mixed = rng.random() < 0.3
    cls0 = int(rng.integers(0, 2))
    for k in range(nf):
        n = int(rng.choice([3000, 4100, 20000, 150000, 700000, 3000000]))
        seg = int(rng.choice([64, 256, 1000]))
        m = max(2, n // seg)
        st = np.linspace(0, n, m)
        style = int(rng.integers(0, 6))
        if style == 0:   sp = 1.0 + 0.01 * np.sin(np.arange(m) * 0.05 + rng.uniform(0, 6))
        elif style == 1: sp = np.full(m, 1.0)
        elif style == 2: sp = 1.0 + 0.03 * np.sin(np.arange(m) * 0.5 + rng.uniform(0, 6))
        elif style == 3: sp = rng.uniform(0.97, 1.0, m)
        elif style == 4: sp = rng.uniform(1.0, 1.03, m)
        else:            sp = 1.0 + 0.02 * np.sign(np.sin(np.arange(m) * 0.3))
        stereo = (cls0 if not mixed else int(rng.integers(0, 2))) == 1
        sig = rng.standard_normal((n, 2) if stereo else n).astype(np.float32)
        if rng.random() < 0.3:
            a = int(rng.integers(0, n // 2)); sig[a:a + n // 7] = 0.0
        if rng.random() < 0.2:
            a = int(rng.integers(0, n // 2)); sig[a:a + 50] = 30.0
        items.append((torch.from_numpy(st).cuda(), torch.from_numpy(sp).cuda(), torch.from_numpy(sig).cuda()))
        host.append((st, sp, sig, n))
    ref = [o.clone() for _, o, _ in R.varispeed_batch_dev(items, 32, group=1)]
    g = int(rng.choice([2, 3, 4, 8]))
    got = [o for _, o, _ in R.varispeed_batch_dev(items, 32, group=g)]
    for k, (o, want) in enumerate(zip(got, ref)):
        if o.shape != want.shape or not torch.equal(o, want):
            print(f"MISMATCH batch {batches} (seed {seed0 + batches}) file {k} group {g}", flush=True)
            sys.exit(1)
    k = int(rng.integers(0, nf))
    st, sp, sig, n = host[k]
    pos, _ = C.speed_to_pos(st, sp, n)
    o = got[k].cpu().numpy()
    for ch in range(sig.shape[1] if sig.ndim == 2 else 1):
        x = sig[:, ch] if sig.ndim == 2 else sig
        y = o[:, ch] if sig.ndim == 2 else o
        want = C.sinc(pos, np.ascontiguousarray(x), 32, threads=8)
        if len(want) != len(y):
            print(f"LENGTH batch {batches} file {k}: {len(y)} vs {len(want)}"); sys.exit(1)
        pk = float(np.abs(want).max()) or 1.0
        worst = max(worst, float(np.abs(y - want).max()) / pk)
    if worst > 5e-6:
        print(f"ORACLE batch {batches} (seed {seed0 + batches}) file {k}: {worst:.3e}"); sys.exit(1)
    batches += 1; files += nf
print(f"group fuzz ok: {batches} batches, {files} files, every merged launch bit-identical to the file-by-file launches; "
      f"worst relative error of the oracle-checked files {worst:.2e}")
