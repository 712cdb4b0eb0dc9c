"""Seeded / closed-form input generators shared by oracle/gen_golden.py and the tests.

Golden fixtures store only the reference's OUTPUTS (plus an input checksum); the
inputs are regenerated from these functions so fixtures stay KB-scale.
"""
import numpy as np


def noise(n, seed=0):
    return np.random.default_rng(seed).standard_normal(n).astype(np.float32)


def sine(n, f, sr, amp=1.0):
    return (amp * np.sin(2 * np.pi * f * np.arange(n) / sr)).astype(np.float32)


def splitmix_uniform(idx, seed=0x5EED):
    """Stateless uniform(-1,1) from a 64-bit hash of the sample index.
    Bit-identical restatement lives in csrc (device) and oracle C."""
    z = (np.asarray(idx, dtype=np.uint64) ^ np.uint64(seed)) + np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0


def bench_signal(start, count, sr, seed=0x5EED):
    """SURVEY 8d primary synthetic signal, evaluated in f64 then cast to f32."""
    n = np.arange(start, start + count, dtype=np.float64)
    x = (0.25 * np.sin(2 * np.pi * 1000.0 * n / sr)
         + 0.25 * np.sin(2 * np.pi * (0.45 * sr / 2) * n / sr)
         + 0.1 * splitmix_uniform(np.arange(start, start + count, dtype=np.uint64), seed))
    return x.astype(np.float32)


def bench_speed_curve(duration_s, sr, hop=256, depth=0.01, rate_hz=0.55, phase=0.7):
    """SURVEY 8d speed curve: [[t_seconds, speed], ...] sampled every ~hop samples
    exactly like util/markers.py:585-588 does (linspace(0, dur, int(dur*sr/hop)))."""
    num = int(duration_s * sr / hop)
    t = np.linspace(0, duration_s, num)
    return np.stack((t, 1 + depth * np.sin(2 * np.pi * rate_hz * t + phase)), axis=-1)


def pilot(n, sr, f0=4000.0, fm_hz=8.0, fm_depth=0.005, noise_db=-60.0, seed=3):
    """FM pilot tone like samples/flutter*.flac ('pilot tone with flutter at 4000 Hz')."""
    t = np.arange(n) / sr
    inst = f0 * (1 + fm_depth * np.sin(2 * np.pi * fm_hz * t))
    phase = 2 * np.pi * np.cumsum(inst) / sr
    x = 0.5 * np.sin(phase) + 10 ** (noise_db / 20) * np.random.default_rng(seed).standard_normal(n)
    return x.astype(np.float32)


def checksum(a):
    a = np.ascontiguousarray(a)
    return float(np.sum(a.astype(np.float64) * (1 + (np.arange(a.size) % 7))))


def detect_input(sr=44100, n=120000):
    """Broadband noise + tone with four smooth-edged dropouts (-20..-26 dB, 8..20 ms) for the detector tests."""
    x = 0.2 * noise(n, 77) + sine(n, 5000.0, sr, 0.2)
    env = np.ones(n)
    for (c, w, depth) in ((20000, 500, 0.05), (47000, 700, 0.08), (80500, 880, 0.05), (101000, 360, 0.1)):
        k = np.arange(c - w, c + w)
        env[k] = np.minimum(env[k], 1 - (1 - depth) * np.hanning(2 * w) ** 0.25)
    return (x * env).astype(np.float32)


def heuristic_input(sr=8000, n=48000, seed=5):
    """Two-channel low-rate test tape for the heuristic dropout repair (dropouts_gui.process_heuristic): band-limited noise
    (15..200 Hz) with three 12-ms dropouts (-26 dB); channel 1 is channel 0 reversed.  The low rate keeps the reference's
    uint16 band arithmetic (dropouts_gui.py:281-282) free of overflow for bands below 128 Hz at fft_size 512."""
    import scipy.signal
    rng = np.random.default_rng(seed)
    x = (0.3 * rng.standard_normal(n)).astype(np.float32)
    sos = scipy.signal.butter(4, [15 / (sr / 2), 200 / (sr / 2)], btype="band", output="sos")
    x = scipy.signal.sosfiltfilt(sos, x).astype(np.float32) * np.float32(4)
    for c in (1.2, 2.9, 4.4):
        i0, w = int(c * sr), int(0.012 * sr)
        x[i0:i0 + w] *= np.float32(0.05)
    return np.stack([x, x[::-1].copy()], axis=1)
