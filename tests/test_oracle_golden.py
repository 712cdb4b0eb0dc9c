"""Pins oracle/oracle_np.py (the CPU restatement) to golden vectors captured from the
real reference by oracle/gen_golden.py.  CPU-only."""
import warnings

import numpy as np
import pytest

import os

import inputs
from oracle import oracle_np as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def relerr(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


STFT_CASES = ["kat1", "bh", "small_vec", "odd_len", "zp2", "zp2_big", "n2048", "n64", "n4096", "hop_eq", "hop_odd"]


@pytest.mark.parametrize("name", STFT_CASES)
def test_stft(golden, name):
    g = golden["stft"]
    n, seed, n_fft, hop, zp = (int(v) for v in g[name + "_cfg"])
    x = inputs.noise(n, seed)
    assert inputs.checksum(x) == float(g[name + "_insum"])
    S = O.stft(x, n_fft, hop, str(g[name + "_win"]), zp)
    assert S.shape == g[name + "_S"].shape == (n_fft * zp // 2 + 1, O.frame_count(n, n_fft, hop))
    assert relerr(S, g[name + "_S"]) < 2e-6
    assert abs(np.abs(S).sum() - float(g[name + "_sum_abs"])) < 1e-5 * float(g[name + "_sum_abs"])


def test_stft_kat_values(golden):
    """SURVEY 8c KAT1/KAT2 literal values."""
    x = inputs.noise(4096, 0)
    S = O.stft(x, 1024, 256, "hann", 1)
    assert abs(S[0, 0] - (-0.19117718935012817)) < 1e-6
    assert abs(S[100, 8] - (-0.9019840955734253 - 0.008981410413980484j)) < 1e-6
    assert abs(np.abs(S).sum() - 4672.315581462234) < 1e-2
    m = O.get_mag(x, 1024, 256, "blackmanharris", 1)
    assert abs(m[5, 3] - 0.44975326198697824) < 1e-6
    assert relerr(m, golden["stft"]["kat2_mag"]) < 2e-6


def test_stft_strided(golden):
    st = np.stack((inputs.noise(3000, 10), inputs.noise(3000, 11)), axis=-1)
    assert relerr(O.stft(st[:, 1], 512, 128, "blackmanharris", 1), golden["stft"]["strided_S"]) < 2e-6


@pytest.mark.parametrize("name", ["rt512", "rt1024", "rt256"])
def test_istft(golden, name):
    g = golden["istft"]
    n, seed, n_fft, hop = (int(v) for v in g[name + "_cfg"])
    x = inputs.noise(n, seed)
    S = O.stft(x, n_fft, hop)
    S2 = S.copy()
    S2[5:40, 3:9] *= 0.25
    assert relerr(O.istft(S, hop_length=hop, length=n), g[name + "_y"]) < 2e-6
    assert relerr(O.istft(S2, hop_length=hop, length=n), g[name + "_ymod"]) < 2e-6
    assert relerr(O.istft(S, hop_length=hop), g[name + "_ynolen"]) < 2e-6
    assert relerr(O.istft(S, hop_length=hop, length=n), x) < 1e-5     # round trip


def test_istft_heal_framing(golden):
    g = golden["istft"]
    x = inputs.noise(5000, 15)
    S = O.stft(O.fix_length(x, len(x) + 256), 512, 32)
    assert tuple(g["heal_shape"]) == S.shape
    assert relerr(O.istft(S, hop_length=32, length=len(x)), g["heal_y"]) < 2e-6


def test_speed_to_pos(golden):
    g = golden["speed_to_pos"]
    n = 8192
    st = np.linspace(0, n, 33)
    sp = 1 + 0.01 * np.sin(2 * np.pi * np.arange(33) / 16 + 0.7)
    pos, trimmed = O.speed_to_pos(st, sp, n)
    assert trimmed and len(pos) == 8191
    assert np.array_equal(pos, g["kat3_pos"])                      # bit-exact float64
    assert abs(pos[4000] - 4001.554017502851) < 1e-9 and abs(pos[-1] - 8191.0160396801675) < 1e-9
    pos, _ = O.speed_to_pos(np.array((0.0, 20000.0)), np.array((0.5, 2.0)), 20000)
    assert np.array_equal(pos, g["ramp_pos"])
    sc = inputs.bench_speed_curve(2.0, 48000)
    pos, _ = O.speed_to_pos(sc[:, 0] * 48000, sc[:, 1], 96000)
    assert np.array_equal(pos, g["bench_pos"])
    pos, _ = O.speed_to_pos(g["wobble_st"], g["wobble_sp"], 30000)
    assert np.array_equal(pos, g["wobble_pos"])
    pos, trimmed = O.speed_to_pos(g["untrimmed_st"], g["untrimmed_sp"], 10000)
    assert not trimmed and len(pos) < int(g["untrimmed_buflen"])
    assert np.array_equal(pos, g["untrimmed_pos"])


def test_sinc(golden):
    g = golden["sinc"]
    gp = golden["speed_to_pos"]
    tol = 3e-7      # f32 output; summation order differs (np.sum pairwise vs axis-sum)
    y = O.sinc_resample(gp["kat3_pos"], inputs.sine(8192, 440, 44100), 32)
    assert relerr(y, g["kat4_y"]) < tol
    assert abs(float(y.sum(dtype=np.float64)) - 24.03584675192542) < 1e-3
    sig = inputs.noise(600, 30)
    y = O.sinc_resample(np.arange(600, dtype=np.float64), sig, 8)
    assert relerr(y, g["ident_y"]) < tol
    assert np.all(np.abs(y[8:] - sig[8:]) < 1e-6) and np.any(np.abs(y[:8] - sig[:8]) > 1e-3)   # quirk 1
    sig = (inputs.sine(20000, 440, 44100, 0.5) + inputs.sine(20000, 21000, 44100, 0.1)).astype(np.float32)
    assert relerr(O.sinc_resample(gp["ramp_pos"], sig, 50), g["ramp_y"]) < tol
    bsig = inputs.bench_signal(0, 96000, 48000)
    yb = O.sinc_resample(gp["bench_pos"], bsig, 32)
    assert relerr(yb, g["bench_y"]) < tol
    assert relerr(yb, g["bench_y_mt"]) < 1e-6           # the reference's own mt == single invariant
    assert relerr(O.sinc_resample(g["tail_pos"], inputs.noise(1000, 31), 16), g["tail_y"]) < tol
    sig = inputs.noise(3000, 32)
    pos = np.cumsum(np.full(2500, 1.013)) - 0.4
    assert relerr(O.sinc_resample(pos, sig, 1), g["nt1_y"]) < tol
    assert relerr(O.sinc_resample(pos, sig, 100), g["nt100_y"]) < tol
    assert relerr(O.sinc_resample(g["down_pos"], sig, 24), g["down_y"]) < tol


def test_filters(golden):
    g = golden["filters"]
    x = inputs.noise(4096, 0).astype(np.float64)
    assert np.allclose(O.butter_bandpass_filter(x, 1000, 4000, 44100, order=3), g["band"], rtol=0, atol=1e-12)
    assert abs(g["band"][100] - (-0.16118795506647626)) < 1e-12          # KAT5
    assert np.allclose(O.butter_bandpass_filter(x, 0, 20, 172.265625, order=3), g["low"], rtol=0, atol=1e-12)
    assert abs(g["low"][100] - (-0.24571490040224575)) < 1e-12           # KAT6
    assert np.allclose(O.butter_bandpass_filter(x, 300, 0, 44100, order=3), g["high"], rtol=0, atol=1e-12)
    assert np.allclose(O.butter_bandpass_filter(x, 500, 2000, 44100), g["band5"], rtol=0, atol=1e-12)
    assert bool(g["pass_is_identity"]) and O.butter_bandpass_filter(x, 0, 0, 44100) is x
    assert np.allclose(O.moving_average(x[:100], 5), g["mavg"])


def test_correlation(golden):
    g = golden["correlation"]
    assert O.parabolic([1, 3, 2], 1) == (1.1666666666666667, 3.0416666666666665) == tuple(g["parabolic"])   # KAT7
    aa = np.sin(np.arange(521) * 1.0)
    bb = np.sin(np.arange(521) * 1.0 + 3)
    assert np.allclose(O.find_delay(aa, bb, window_name="hann"), g["find_delay"])
    assert np.allclose(O.xcorr(inputs.noise(200, 40).astype(np.float64), inputs.noise(200, 41).astype(np.float64),
                               mode="same"), g["xcorr_same"])


def test_trackers(golden):
    g = golden["trackers"]
    sr, n, n_fft, hop = (int(v) for v in g["cfg"])
    x = inputs.pilot(n, sr)
    spec = O.get_mag(x, n_fft, hop, "blackmanharris", 1)
    trail = [(0.2, 4000.0), (1.3, 4000.0)]
    for name, key in (("Peak", "peak"), ("Peak Track", "peak_track"), ("Center of Gravity", "center_of_gravity"),
                      ("Correlation", "correlation"), ("Freehand Draw", "freehand_draw")):
        t, f = O.TRACKERS[name](spec, list(trail), n_fft, hop, sr, 0.5)
        assert np.array_equal(t, g[key + "_times"]), name
        assert relerr(f, g[key + "_freqs"]) < 1e-6, name
    t, f = O.track_zero_crossing(spec, x[:, None], list(trail), n_fft, hop, sr, 0.5)
    assert np.array_equal(t, g["zero_crossing_times"]) and relerr(f, g["zero_crossing_freqs"]) < 1e-9
    t, f = O.track_peak(spec, [(1.2, 4030.0), (0.1, 3980.0), (0.6, 4010.0)], n_fft, hop, sr, 2.0)
    assert np.array_equal(t, g["peak2_times"]) and relerr(f, g["peak2_freqs"]) < 1e-6
    # sanity: the pilot's FM is recovered (8 Hz, +-0.5 %)
    assert abs(np.mean(g["peak_freqs"]) - 4000) < 5 and 10 < np.ptp(g["peak_freqs"]) < 60


def test_pipeline(golden):
    g = golden["pipeline"]
    sr, n, n_fft, hop = (int(v) for v in g["cfg"])
    x = inputs.pilot(n, sr)
    spec = O.get_mag(x, n_fft, hop, "blackmanharris", 1)
    t, f = O.track_peak(spec, [(0.05, 4000.0), (1.45, 4000.0)], n_fft, hop, sr, 0.5)
    assert np.array_equal(t, g["track_times"]) and relerr(f, g["track_freqs"]) < 1e-6
    curve = O.master_speed_curve([(t, O.trace_to_speed(f))], n / sr, sr, hop, bands=(0, 20))
    assert curve.shape == g["curve"].shape and relerr(curve[:, 1], g["curve"][:, 1]) < 1e-7
    pos, _ = O.speed_to_pos(curve[:, 0] * sr, curve[:, 1], n)
    assert len(pos) == len(g["pos"]) and np.max(np.abs(pos - g["pos"])) < 1e-5
    y = O.sinc_resample(g["pos"], x, 32)
    assert relerr(y, g["y"]) < 3e-7


def test_config3_conditioning(golden):
    """Why the END-TO-END config-3 bound is not the per-stage 1e-5 (tests/test_hip_parity.py::P0_BACKEND_SPREAD): the
    flow amplifies.  (a) The reference's interpolant centres its window on round(p) (util/resampling.py:60-75), so a
    position moved by 1e-5 samples across a half-integer changes the output by more than 1e-5 of peak, and even
    without a flip the output follows the position with slope ~0.5/sample.  (b) Positions are a running sum over the
    file: the same flow on the float64 transform numpy < 2 computes (rfft upcasts) instead of numpy >= 2's float32
    moves the tracked frequencies by ~1e-9 and the positions by ~1000 times that.  tools/p0_sensitivity.py prints
    the full table, torch.stft backend and flutter_192.flac included (profiles/r02_p0_sensitivity.txt)."""
    import scipy.signal
    g = golden["pipeline"]
    sr, n, n_fft, hop = (int(v) for v in g["cfg"])
    x = inputs.pilot(n, sr)
    y0 = O.sinc_resample(g["pos"], x, 32)
    y1 = O.sinc_resample(g["pos"] + 1e-5, x, 32)
    e = np.abs(y1 - y0) / np.max(np.abs(y0))
    flips = np.rint(g["pos"] + 1e-5) != np.rint(g["pos"])
    assert flips.sum() >= 1 and e[flips].max() > 1e-5 and e[~flips].max() < 1e-5
    assert 2e-6 < np.median(e) < 1e-5                                 # slope: ~0.37 of peak per sample at the median
    win = scipy.signal.get_window("blackmanharris", n_fft).astype(np.float32)
    xp = np.pad(x, n_fft // 2, mode="reflect")
    idx = np.arange((len(xp) - n_fft) // hop + 1)[:, None] * hop + np.arange(n_fft)[None, :]
    frames = (win[None, :] * xp[idx]).astype(np.float32)
    mag64 = np.abs(np.fft.rfft(frames.astype(np.float64), axis=1).T / np.sqrt(n_fft)) + 1e-7
    t, f = O.track_peak(mag64, [(0.05, 4000.0), (1.45, 4000.0)], n_fft, hop, sr, 0.5)
    curve = O.master_speed_curve([(t, O.trace_to_speed(f))], n / sr, sr, hop, bands=(0, 20))
    pos, _ = O.speed_to_pos(curve[:, 0] * sr, curve[:, 1], n)
    df, dp = relerr(f, g["track_freqs"]), np.max(np.abs(pos - g["pos"]))
    assert 1e-10 < df < 1e-8 and len(pos) == len(g["pos"]) and dp > 300 * df


def test_piptrack_restatement_by_hand():
    """oracle piptrack (librosa's published algorithm, parity unpinned) on columns whose answer is computable by hand:
    a parabola sampled at integers has its vertex recovered exactly; thresholding, the frequency mask, the one-sided local
    maximum rule (x[k] > x[k-1] and x[k] >= x[k+1]) and the never-selected edge bins."""
    sr, n_fft = 8000, 16                        # bins 0..8, 500 Hz apart
    k = np.arange(9.0)
    col0 = np.maximum(0.0, 10.0 - 2.0 * (k - 3.3) ** 2)               # vertex at bin 3.3 (1650 Hz), height 10
    col1 = np.array([5.0, 1.0, 4.0, 4.0, 1.0, 0.2, 0.3, 0.2, 9.0])    # plateau over bins 2-3
    col2 = np.zeros(9)
    S = np.stack((col0, col1, col2), axis=1)
    p, m = O.piptrack(S, sr, n_fft, fmin=400.0, fmax=3900.0, threshold=0.15)
    assert p.shape == S.shape and np.count_nonzero(p[:, 0]) == 1
    assert abs(p[3, 0] - 3.3 * 500.0) < 1e-9 and abs(m[3, 0] - 10.0) < 1e-9
    # column 1: bin 0 (left edge) and bin 8 (right edge, = sr/2, outside fmax) never count; the plateau's first bin wins
    # (4 > 1 on the left, 4 >= 4 on the right), its second does not (4 > 4 fails); bin 6 (0.3) is under the threshold 1.35
    assert list(np.nonzero(p[:, 1])[0]) == [2] and p[2, 1] == (2 + 0.5) * 500.0 and m[2, 1] == 4.0 + 0.5 * 1.5 * 0.5
    assert not p[:, 2].any() and not m[:, 2].any()


def test_linear_and_lag(golden):
    g = golden["linear_lag"]
    sig = inputs.noise(5000, 50)
    pos = O.lag_to_positions(g["lag"], int(g["sr"]), len(sig))
    assert np.array_equal(pos, g["pos"])
    assert np.array_equal(O.linear_resample(pos, sig), g["lin"])


# ----------------------------------------------------------------- plain-C oracle (oracle/par_oracle.c)

def test_c_oracle_speed_to_pos(golden):
    from oracle import oracle_c as C
    g = golden["speed_to_pos"]
    n = 8192
    st = np.linspace(0, n, 33)
    sp = 1 + 0.01 * np.sin(2 * np.pi * np.arange(33) / 16 + 0.7)
    pos, trimmed = C.speed_to_pos(st, sp, n)
    assert trimmed and np.array_equal(pos, g["kat3_pos"])
    pos, _ = C.speed_to_pos(np.array((0.0, 20000.0)), np.array((0.5, 2.0)), 20000)
    assert np.array_equal(pos, g["ramp_pos"])
    sc = inputs.bench_speed_curve(2.0, 48000)
    pos, _ = C.speed_to_pos(sc[:, 0] * 48000, sc[:, 1], 96000)
    assert np.array_equal(pos, g["bench_pos"])
    pos, _ = C.speed_to_pos(g["wobble_st"], g["wobble_sp"], 30000)
    assert np.array_equal(pos, g["wobble_pos"])
    pos, trimmed = C.speed_to_pos(g["untrimmed_st"], g["untrimmed_sp"], 10000)
    assert not trimmed and np.array_equal(pos, g["untrimmed_pos"])


def test_c_oracle_speed_to_pos_windows(golden):
    """oracle_speed_to_pos_windows (the windowed, threaded form bench.py's parity check and the full-size tests use) against the
    reference's own positions (golden fixtures) and against oracle_speed_to_pos, bit for bit -- including windows over segment
    borders, over the trim, beyond the end, and curves that end untrimmed"""
    from oracle import oracle_c as C
    g = golden["speed_to_pos"]
    n = 8192
    st = np.linspace(0, n, 33)
    sp = 1 + 0.01 * np.sin(2 * np.pi * np.arange(33) / 16 + 0.7)
    sc = inputs.bench_speed_curve(2.0, 48000)
    cases = ((st, sp, n, g["kat3_pos"], True), (np.array((0.0, 20000.0)), np.array((0.5, 2.0)), 20000, g["ramp_pos"], None),
             (sc[:, 0] * 48000, sc[:, 1], 96000, g["bench_pos"], None), (g["wobble_st"], g["wobble_sp"], 30000, g["wobble_pos"], None),
             (g["untrimmed_st"], g["untrimmed_sp"], 10000, g["untrimmed_pos"], False))
    for st_, sp_, n_, want, trimmed in cases:
        for width, threads in ((1, 1), (300, 3), (len(want) + 7, 16)):
            starts = sorted({0, 1, 255, 256, 257, len(want) // 3, max(0, len(want) - width), len(want) - 1, len(want), len(want) + 11})
            win, len_out, tr = C.speed_to_pos_windows(st_, sp_, n_, starts, width, threads=threads)
            assert len_out == len(want) and (trimmed is None or tr == trimmed)
            for w, s0 in enumerate(starts):
                ref = np.full(width, np.nan)
                k = max(0, min(width, len(want) - s0))
                ref[:k] = want[s0:s0 + k]
                assert np.array_equal(win[w], ref, equal_nan=True), (len(want), width, s0)


def test_c_oracle_sinc(golden):
    from oracle import oracle_c as C
    g = golden["sinc"]
    gp = golden["speed_to_pos"]
    tol = 3e-7
    assert relerr(C.sinc(gp["kat3_pos"], inputs.sine(8192, 440, 44100), 32), g["kat4_y"]) < tol
    assert relerr(C.sinc(np.arange(600, dtype=np.float64), inputs.noise(600, 30), 8), g["ident_y"]) < tol
    sig = (inputs.sine(20000, 440, 44100, 0.5) + inputs.sine(20000, 21000, 44100, 0.1)).astype(np.float32)
    assert relerr(C.sinc(gp["ramp_pos"], sig, 50), g["ramp_y"]) < tol
    bsig = inputs.bench_signal(0, 96000, 48000)
    assert relerr(C.sinc(gp["bench_pos"], bsig, 32), g["bench_y"]) < tol
    assert relerr(C.sinc(gp["bench_pos"], bsig, 32, threads=5), g["bench_y"]) < tol     # mt == single
    assert relerr(C.sinc(g["tail_pos"], inputs.noise(1000, 31), 16), g["tail_y"]) < tol
    sig = inputs.noise(3000, 32)
    pos = np.cumsum(np.full(2500, 1.013)) - 0.4
    assert relerr(C.sinc(pos, sig, 1), g["nt1_y"]) < tol
    assert relerr(C.sinc(pos, sig, 100), g["nt100_y"]) < tol
    assert relerr(C.sinc(g["down_pos"], sig, 24), g["down_y"]) < tol


@pytest.mark.parametrize("name", ["kat1", "small_vec", "odd_len", "zp2", "n2048", "n64", "hop_odd"])
def test_c_oracle_stft(golden, name):
    from oracle import oracle_c as C
    import scipy.signal
    g = golden["stft"]
    n, seed, n_fft, hop, zp = (int(v) for v in g[name + "_cfg"])
    win = scipy.signal.get_window(str(g[name + "_win"]), n_fft).astype(np.float32)
    S = C.stft(inputs.noise(n, seed), n_fft, hop, win, zp, mode=0)
    assert relerr(S, g[name + "_S"]) < 2e-6
    m = C.stft(inputs.noise(n, seed), n_fft, hop, win, zp, mode=1)
    assert relerr(m, np.abs(g[name + "_S"]) + 1e-7) < 2e-6


def test_c_oracle_synth_matches_numpy_generators():
    from oracle import oracle_c as C
    a = C.synth_signal(12345, 5000, 192000.0)
    b = inputs.bench_signal(12345, 5000, 192000.0)
    assert np.max(np.abs(a - b)) <= 6e-8            # last-ulp libm differences before the f32 cast
    sc = inputs.bench_speed_curve(3.0, 48000)
    st, sp = C.synth_curve(len(sc), 3.0, 48000.0)
    assert np.allclose(st, sc[:, 0] * 48000, rtol=0, atol=1e-7) and np.allclose(sp, sc[:, 1], rtol=0, atol=1e-15)


def heal_input(sr):
    x = (inputs.sine(30000, 1500.0, sr, 0.4) + 0.05 * inputs.noise(30000, 60)).astype(np.float32)
    x[9000:9300] *= 0.05
    x[20000:20500] *= 0.1
    return x


def test_heal_dropouts_config4(golden):
    g = golden["heal"]
    sr = int(g["sr"])
    x = heal_input(sr)
    y = O.heal_dropouts(x, sr, [tuple(m) for m in g["marks"]], 512, 32)
    assert relerr(y[:, 0], g["y"]) < 2e-6
    assert np.abs(y[9000:9300, 0]).mean() > 4 * np.abs(x[9000:9300]).mean()      # the dropout was lifted
    assert relerr(y[:8000, 0], x[:8000]) < 1e-4                                   # untouched region survives the round trip


def test_c_oracle_buffer_bound_is_numpys():
    """The reference raises when a segment does not fit its end_guess buffer (util/resampling.py:108-109, :127); the C
    oracle must size that buffer with numpy's own pairwise mean, or it would accept curves the reference refuses
    (found by tools/fuzz_resampler.py)."""
    from oracle import oracle_c as C
    L = C.lib()
    rng = np.random.default_rng(5)
    for _ in range(3000):
        m = int(rng.integers(2, 4000))
        sp = rng.uniform(0.3, 3.0, m)
        st = np.linspace(0, rng.uniform(10, 1e6), m)
        assert int(L.oracle_end_guess(C._p(st), C._p(sp), m)) == int(np.mean(sp) * (st[-1] - st[0]) * 1.01)
    # a curve whose last segment overflows the buffer before the trim test: both restatements refuse it
    rng = np.random.default_rng(2690 + 1436)
    n = int(rng.choice([3000, 20000, 150000, 700000]))
    rng.choice(16)
    seg = int(rng.choice([16, 64, 256, 1000, 5000]))
    m = max(2, n // seg)
    st = np.linspace(0, n, m)
    rng.integers(0, 5)
    sp = rng.uniform(0.5, 2.0, m)
    with pytest.raises(ValueError):
        O.speed_to_pos(st, sp, n)
    with pytest.raises(ValueError):
        C.speed_to_pos(st, sp, n)


def test_dropout_detector(golden):
    """dropout_healer_gui.py:185-242 restated on the oracle's get_mag vs the fixture made on the reference's."""
    g = golden["detect"]
    sr, fft, hop, t0, t1, fl, fu = g["args"]
    m = O.get_mag(inputs.detect_input(int(sr)), int(fft), int(hop), "blackmanharris", 1)
    vol, fb = O.band_volume_db(m, int(sr), int(fft), int(hop), t0, t1, fl, fu)
    assert fb == int(g["frame_b"]) and np.abs(vol - g["vol"]).max() < 1e-4           # dB
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        found = O.detect_dropouts(m, int(sr), int(fft), int(hop), t0, t1, fl, fu, 20, 5)
    got = np.array([(a[0], a[1], b[0], b[1]) for a, b in found])
    assert got.shape == g["found"].shape and np.abs(got - g["found"]).max() < 1e-9
    centres = (got[:, 0] + got[:, 2]) / 2
    for c in (20000, 47000, 80500, 101000):                                          # every planted dropout is found
        assert np.abs(centres - c / sr).min() < 0.004


# ------------------------------------------------- BASELINE configs 1 and 3 on the reference's sample files
def grid(a, k=97):
    return np.asarray(a).ravel()[::k]


def test_flac_decoder_reads_the_reference_samples():
    """tests/golden/flutter*.flac are DATA files of the reference (its demo inputs).  The decoder verifies the
    STREAMINFO MD5 of the decoded PCM itself; sizes/rates are the ones SURVEY 8c recorded."""
    from pyaudiorestoration_amd import io_ops
    x, sr, ch = io_ops.read_file(os.path.join(GOLD, "flutter.flac"))
    assert x.shape == (186291, 1) and x.dtype == np.float32 and sr == 44100 and ch == 1
    x, sr, ch = io_ops.read_file(os.path.join(GOLD, "flutter_192.flac"))
    assert x.shape == (811063, 1) and sr == 192000 and ch == 1


def test_config1_flutter_stft_plumbing(golden):
    from pyaudiorestoration_amd import io_ops
    g = golden["samples"]
    x, sr, _ = io_ops.read_file(os.path.join(GOLD, "flutter.flac"))
    m = O.get_mag(x[:, 0], 1024, 256, "hann", 1)
    assert m.shape == tuple(g["c1_shape"]) == (513, 728)
    assert relerr(grid(m), g["c1_grid"]) < 2e-6 and abs(m.sum() - float(g["c1_sum"])) < 1e-5 * float(g["c1_sum"])


def test_config3_flutter192_pipeline(golden):
    from pyaudiorestoration_amd import io_ops
    from oracle import oracle_c as C
    g = golden["samples"]
    x, sr, _ = io_ops.read_file(os.path.join(GOLD, "flutter_192.flac"))
    spec = O.get_mag(x[:, 0], 1024, 256, "blackmanharris", 1)
    assert spec.shape == tuple(g["c3_shape"]) == (513, 3169) and relerr(grid(spec), g["c3_grid"]) < 2e-6
    t, f = O.track_peak(spec, [(0.2, 4000.0), (4.0, 4000.0)], 1024, 256, sr, 0.5)
    assert np.array_equal(t, g["c3_track_times"]) and relerr(f, g["c3_track_freqs"]) < 1e-6
    curve = O.master_speed_curve([(t, O.trace_to_speed(f))], len(x) / sr, sr, 256, bands=(0, 20))
    assert relerr(curve[:, 1], g["c3_curve"][:, 1]) < 1e-7
    pos, _ = C.speed_to_pos(g["c3_curve"][:, 0] * sr, g["c3_curve"][:, 1], len(x))
    assert len(pos) == int(g["c3_len_pos"]) and np.array_equal(pos[::1009], g["c3_pos_grid"])
    y = C.sinc(pos, x[:, 0], 32, threads=8)
    assert relerr(y[g["c3_sel"]], g["c3_y_sel"]) < 3e-7


def test_spd_project_master_curves(golden):
    """r04: two saved pyrespeeder projects (tests/golden/flutter_192_{traces,reg}.spd, written by the reference's own
    save_json from TraceLine / RegLine.to_cfg rows) and the curves the reference's OWN MasterSpeedLine.update /
    MasterRegLine.update / Canvas.get_speed_curve made of them (oracle/ref_gui.py drives them headless): the oracle's
    restatement must reproduce them, and the C oracle their positions and output."""
    import json
    from pyaudiorestoration_amd import io_ops
    from oracle import oracle_c as C
    g = golden["spd"]
    x, sr, _ = io_ops.read_file(os.path.join(GOLD, "flutter_192.flac"))
    for tag in ("traces", "reg"):
        cfg = json.load(open(os.path.join(GOLD, f"flutter_192_{tag}.spd")))
        assert len(cfg["lines"]) == 2 and len(cfg["regs"]) == (1 if tag == "reg" else 0)
        curve = O.project_speed_curve(cfg, len(x) / sr, sr)
        assert curve.shape == g[tag + "_curve"].shape and relerr(curve[:, 1], g[tag + "_curve"][:, 1]) < 1e-12
        assert np.array_equal(curve[:, 0], g[tag + "_curve"][:, 0])
        pos, _ = C.speed_to_pos(g[tag + "_curve"][:, 0] * sr, g[tag + "_curve"][:, 1], len(x))
        assert len(pos) == int(g[tag + "_len_pos"]) and np.array_equal(pos[::1009], g[tag + "_pos_grid"])
        y = C.sinc(pos, x[:, 0], 32, threads=8)
        assert relerr(y[g[tag + "_sel"]], g[tag + "_y_sel"]) < 3e-7
    # the regressed curve is the one Canvas.get_speed_curve hands out when a project holds regressions (1.5 x the fitted sine)
    amp = float(g["reg"][2])
    assert abs(np.max(np.log2(g["reg_curve"][:, 1])) - 1.5 * abs(amp)) < 1e-3 * abs(amp)
    assert relerr(np.log2(g["reg_curve"][:, 1]), g["master_reg"][:, 1]) < 1e-12


def test_heuristic_repair_oracle_matches_the_reference():
    """oracle_np.heal_heuristic against dropouts_gui.MainWindow.process_heuristic itself (fixture generated by
    oracle/gen_golden.py through oracle/ref_gui.heuristic_through_reference): a 8 kHz two-channel tape with low bands (no uint16
    overflow in the reference's band arithmetic) and the GUI's defaults on dropouts_sample.flac (where, under numpy >= 2, that
    arithmetic wraps: the bands sit in bins 0..1 -- reproduced)."""
    import inputs
    from oracle import oracle_np as O
    from pyaudiorestoration_amd import io_ops
    g = np.load(os.path.join(GOLD, "heuristic.npz"))
    sig = inputs.heuristic_input()
    assert inputs.checksum(sig.ravel()) == float(g["low_in_sum"])
    sr, fft, hop, mw, ms, nb, bf, fu, fl = g["low_params"]
    y = O.heal_heuristic(sig, int(sr), int(fft), int(hop), float(mw), float(ms), int(nb), float(bf), float(fu), float(fl))
    assert y.dtype == np.float32 and np.array_equal(y, g["low"])
    assert float(np.max(np.abs(y - sig))) > 0.1                      # the repair did something
    assert O.heuristic_bins(np.uint16(3000), np.uint16(6000), 512, 44100) == (0, 1)      # the wrap, stated
    x, sr_d, _ = io_ops.read_flac(os.path.join(GOLD, "dropouts_sample.flac"))
    y = O.heal_heuristic(x, sr_d, 512, 64)
    assert np.array_equal(y[::3], g["gui_every3"]) and abs(float(np.max(np.abs(y - x))) - float(g["gui_changed"])) < 1e-9
