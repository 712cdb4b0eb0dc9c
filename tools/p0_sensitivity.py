#!/usr/bin/env python3
"""How well-conditioned is the config-3 flow (STFT -> Peak tracker -> speed curve -> positions -> sinc resample)?
Runs the oracle's restatement of the flow with the spectrogram computed by each of the transforms the reference's own
backend chain can end up with (util/fourier.py:67-75: torch.stft float32 first, pyfftw float32, numpy rfft -- float32
under numpy >= 2, float64 before) and prints how far the tracked frequencies, the curve, the positions and the resampled
output move between them, relative to the numpy >= 2 row (what tests/golden/ was generated with).
    python tools/p0_sensitivity.py [--c3] [--gpu]
--c3: the reference's flutter_192.flac (BASELINE config 3) instead of the 1.5-s pilot of tests/golden/pipeline.npz
--gpu: add this build's pipeline.respeed (K_stft -> K_track -> plan -> K_sinc) as a row (needs the GPU)
CPU rows use oracle/ only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scipy.signal
import torch

import inputs
from oracle import oracle_np as O

GOLDEN = "numpy rfft float32 (numpy >= 2: the golden fixtures)"


def windows_of(n_pos):
    """The output windows the c3 fixture keeps (oracle/gen_golden.py): head, middle, tail."""
    return [(0, 3000), (400000, 403000), (n_pos - 3000, n_pos)] if n_pos > 500000 else [(0, n_pos)]


def flow(mag, x, sr, n_fft, hop, trail):
    times, freqs = O.track_peak(mag, list(trail), n_fft, hop, sr, 0.5)
    curve = O.master_speed_curve([(times, O.trace_to_speed(freqs))], len(x) / sr, sr, hop, (0, 20))
    pos = O.speed_to_pos(curve[:, 0] * sr, curve[:, 1], len(x))[0]
    y = np.concatenate([O.sinc_resample(pos[a:min(b + 1, len(pos))], x, 32)[:b - a] for a, b in windows_of(len(pos))])
    return freqs, curve, pos, y


def spectrograms(x, n_fft, hop):
    win = scipy.signal.get_window("blackmanharris", n_fft).astype(np.float32)
    xp = np.pad(x, n_fft // 2, mode="reflect")
    nf = (len(xp) - n_fft) // hop + 1
    out = {GOLDEN: O.get_mag(x, n_fft, hop, "blackmanharris")}
    m64 = np.empty((n_fft // 2 + 1, nf))
    for f0 in range(0, nf, 4096):                               # numpy < 2: rfft upcasts the float32 frames
        idx = np.arange(f0, min(f0 + 4096, nf))[:, None] * hop + np.arange(n_fft)[None, :]
        fr = (win[None, :] * xp[idx]).astype(np.float32)
        m64[:, f0:f0 + 4096] = np.abs(np.fft.rfft(fr.astype(np.float64), axis=1).T / np.sqrt(n_fft)) + 1e-7
    out["numpy rfft float64 (numpy < 2 upcasts)"] = m64
    S = torch.stft(torch.as_tensor(x, dtype=torch.float32), n_fft, hop_length=hop, window=torch.as_tensor(win),
                   win_length=n_fft, center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    S /= np.sqrt(n_fft)
    out["torch.stft float32 (the reference's FIRST choice; here on the CPU)"] = np.abs(S.numpy()) + 1e-7
    rng = np.random.default_rng(0)
    base = out[GOLDEN]
    out["golden spectrogram x (1 + 6e-8 U(-1,1)): one float32 rounding"] = base * (1 + 6e-8 * rng.uniform(-1, 1, base.shape))
    return out


def load(c3):
    if c3:
        from pyaudiorestoration_amd import io_ops
        x, sr, _ = io_ops.read_file(os.path.join(ROOT, "tests", "golden", "flutter_192.flac"))
        return np.ascontiguousarray(x[:, 0]), sr, 1024, 256, [(0.2, 4000.0), (4.0, 4000.0)]
    g = np.load(os.path.join(ROOT, "tests", "golden", "pipeline.npz"))
    sr, n, n_fft, hop = (int(v) for v in g["cfg"])
    return inputs.pilot(n, sr), sr, n_fft, hop, [(0.05, 4000.0), (1.45, 4000.0)]


def table(c3=False, gpu=False, out=sys.stdout):
    x, sr, n_fft, hop, trail = load(c3)
    res = {k: flow(m, x, sr, n_fft, hop, trail) for k, m in spectrograms(x, n_fft, hop).items()}
    if gpu:
        from pyaudiorestoration_amd import pipeline
        r = pipeline.respeed(x, sr, trail, n_fft, hop, 1, "Peak", 0.5, (0, 20), 32)
        pos, y = r["positions"].cpu().numpy(), r["output"].cpu().numpy()[:, 0]
        res["THIS BUILD: K_stft -> K_track (band re-read from the signal, float64) -> K_sinc"] = (
            r["freqs"], r["speed_curve"], pos, np.concatenate([y[a:b] for a, b in windows_of(len(pos))]))
    ref = res[GOLDEN]
    print(f"{'config 3 (flutter_192.flac)' if c3 else 'pilot (tests/golden/pipeline.npz)'}: {len(x)} samples, sr {sr}", file=out)
    print(f"{'spectrogram from':74s} {'freqs rel':>9s} {'curve rel':>9s} {'pos abs':>9s} {'out max':>9s} {'out median':>10s} {'>1e-5':>6s} "
          f"{'>1e-5 where round(pos) agrees':>30s}", file=out)
    rows = {}
    for k, (f, c, p, y) in res.items():
        assert len(p) == len(ref[2]), "position count differs"
        e = np.abs(y - ref[3]) / np.max(np.abs(ref[3]))
        flips = np.concatenate([(np.rint(p) != np.rint(ref[2]))[a:b] for a, b in windows_of(len(p))])
        rows[k] = (np.max(np.abs(f - ref[0]) / ref[0]), np.max(np.abs(c[:, 1] - ref[1][:, 1]) / ref[1][:, 1]),
                   np.max(np.abs(p - ref[2])), e.max(), np.median(e), int((e > 1e-5).sum()), int((e[~flips] > 1e-5).sum()))
        print(f"{k:74s} {rows[k][0]:9.2e} {rows[k][1]:9.2e} {rows[k][2]:9.2e} {rows[k][3]:9.2e} {rows[k][4]:10.2e} {rows[k][5]:6d} "
              f"{rows[k][6]:30d}", file=out)
    return rows


if __name__ == "__main__":
    table("--c3" in sys.argv, "--gpu" in sys.argv)
