"""GPU parity tests: the HIP path (through the Python mirror -> ctypes -> C ABI) against
 (1) golden vectors captured from the real reference, (2) the pinned CPU oracle on seeded inputs,
 (3) size-independent properties at larger sizes.
Tolerances: float outputs within 1e-5 relative (norm-wise) -- most are ~1e-6;
positions, lengths and frame counts exact."""
import ctypes
import numpy as np
import pytest
import scipy.signal

import inputs

pytestmark = pytest.mark.gpu

TOL = 1e-5


# End-to-end bound of the config-3 flow (STFT -> tracker -> curve -> positions -> resample).  The stage tests hold TOL;
# the chain cannot quite: the reference's own STFT backends (torch.stft float32 first, numpy last -- util/fourier.py:67-75)
# move its output by 1.4e-5 (numpy < 2 vs >= 2) to 1.1e-4 (torch on one x86 host; 8e-6 on another) on flutter_192.flac
# against the numpy >= 2 run the fixtures hold (profiles/r03_p0_sensitivity.txt; tools/p0_sensitivity.py reproduces it).
# Since r03 the Peak tracker reads its band from the signal in float64 and this build sits exactly on the float64-rfft
# row of that table: 1.42e-5 on flutter_192.flac with 6 of 811 k samples over 1e-5 -- position DRIFT (5.6e-5 samples after
# 8e5: the fixture's own float32 FFT noise in the tracked frequencies, summed over the file), not half-integer flips --
# and 2.0e-6 on the 1.5-s pilot (r02: 3.7e-5 / 2.9e-5 with float32 band magnitudes).
P0_BACKEND_SPREAD = 2e-5      # documented bound of the MAX only; what the end-to-end tests assert is chain_parity() below


def chain_parity(y, ref, peak, what, max_over, q=0.9999):
    """The end-to-end contract of the config-3 chain: stage-wise 1e-5 (asserted where the stages are tested); over
    the chain, at most `max_over` of the compared samples beyond 1e-5 of the peak, the q-quantile of the differences inside 1e-5,
    and the largest one inside the spread the reference's own STFT backends show on this file.  A regression that moved every
    sample by 1.9e-5 passed the old max-norm bound of 2e-5; it fails this."""
    d = np.abs(np.asarray(y, dtype=np.float64) - np.asarray(ref, dtype=np.float64)) / float(peak)
    over = int((d > TOL).sum())
    assert over <= max_over, (what, "samples beyond 1e-5 of the peak", over, len(d), float(d.max()))
    assert float(np.quantile(d, q)) < TOL, (what, q, float(np.quantile(d, q)))
    assert float(d.max()) < P0_BACKEND_SPREAD, (what, float(d.max()))
    return over, float(d.max())


def relerr(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    # (float64 scalars: under numpy 2 a float32 numerator turns the python float 1e-300 into float32 0 -- an all-zero float32
    # reference, e.g. a window beyond the signal's end, then read 0 / 0 = nan instead of 0)
    return float(np.float64(np.max(np.abs(a - b))) / max(float(np.max(np.abs(b))), 1e-300))


def block_relerr(a, b, block=4096, floor_db=-80.0):
    """max over `block`-sample blocks of max|a - b| / max|b| WITHIN the block; blocks whose peak lies below `floor_db` dBFS of the
    file's peak are skipped.  The norm-wise `relerr` lets a loud passage vouch for a quiet one; this does not."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    k = (len(b) // block) * block
    if k == 0:
        return relerr(a, b)
    pk = np.max(np.abs(b))
    bw = np.abs(b[:k]).reshape(-1, block).max(1)
    be = np.abs(a[:k] - b[:k]).reshape(-1, block).max(1)
    keep = bw > pk * 10.0 ** (floor_db / 20.0)
    return float(np.max(be[keep] / bw[keep])) if keep.any() else 0.0


# (r06: 2e-6 -> 4e-6.  The streaming kernel's (e1 d1) filters no longer take the signal's lo image -- two matrix-core
# instructions per pass -- which moves its error on noise from 7e-7 to 1.5e-6 of the peak, 2.5e-6 on the worst tones; against the
# ORACLE every streaming test holds the contract's 1e-5, this bound only says how far two HIP paths may sit apart)
FUSED_TOL = 4e-6


def same_resample(fused, via_pos):
    """Fused K_sinc against the position-array K_sinc on the oracle-exact float64 positions.  Every window centre
    rint(p) is the same (near-ties are redone with the reference's own arithmetic); the sub-sample shift comes from the
    closed form, which is accurate to 1e-10 where the reference's own position carries half an ulp of rounding (6e-8 at
    7e8), so the two outputs agree to a few 1e-7 of the peak, not bit for bit."""
    a, b = fused.float(), via_pos.float()
    if a.shape != b.shape:
        return False
    return bool(((a - b).abs().max() <= FUSED_TOL * b.abs().max()).item()) and bool((a.isnan() == b.isnan()).all().item())



def oracle_spot(out_t, pos, sig_t, NT, windows=5, width=1200, sig_stride=1, tol=TOL):
    """A few windows of a device output against the C oracle's sinc at the given float64 positions (numpy, or a device tensor
    that another assertion holds bit-equal to the oracle's) -- VERDICT r05 1b: tests that compared two HIP paths with each other
    (same_resample) now also say what the numbers ARE.  Windows: spread, first and last."""
    from oracle import oracle_c as C
    pos = pos.cpu().numpy() if hasattr(pos, "cpu") else np.asarray(pos)
    n_out = len(pos)
    sig = (sig_t.cpu().numpy() if hasattr(sig_t, "cpu") else np.asarray(sig_t)).reshape(-1)[::sig_stride]
    out = out_t.reshape(-1) if hasattr(out_t, "reshape") else out_t
    width = min(width, n_out - 1)
    worst = 0.0
    for i in sorted({int(v) for v in np.linspace(0, n_out - width - 1, windows)}):
        p = pos[i:i + width + 1]
        finite = np.isfinite(p)
        lo = int(max(0, np.floor(np.min(p[finite])) - 200)) if finite.any() else 0
        hi = int(min(len(sig), np.ceil(np.max(p[finite])) + 200)) if finite.any() else len(sig)
        if lo == 0 or hi - lo > 4_000_000 or lo > len(sig) - 4096:      # (window-local coordinates only where they are safe and pay:
            # the leading edge and windows at or beyond the signal's end -- an empty or clipped slice -- keep absolute ones)
            ref = C.sinc(p, sig, NT)[:width]
        else:
            ref = C.sinc(p - lo, sig[lo:hi], NT)[:width]
        got = out[i:i + width]
        got = got.cpu().numpy() if hasattr(got, "cpu") else np.asarray(got)
        e = relerr(got, ref)
        worst = max(worst, e)
        assert e < tol, (i, e)
    return worst


@pytest.fixture(scope="module")
def par():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from pyaudiorestoration_amd import _lib, correlation, fourier, resampling, wow_detection, filters, pipeline
    assert _lib.lib().par_device_count() >= 1

    class P:
        pass
    p = P()
    p.fourier, p.resampling, p.wow, p.filters, p.pipeline, p.torch = fourier, resampling, wow_detection, filters, pipeline, torch
    p.correlation = correlation
    return p


# ------------------------------------------------------------------------------------ STFT
STFT_CASES = ["kat1", "bh", "small_vec", "odd_len", "zp2", "zp2_big", "n2048", "n64", "n4096", "hop_eq", "hop_odd"]


@pytest.mark.parametrize("name", STFT_CASES)
def test_stft_golden(par, golden, name):
    g = golden["stft"]
    n, seed, n_fft, hop, zp = (int(v) for v in g[name + "_cfg"])
    x = inputs.noise(n, seed)
    S = par.fourier.stft(x, n_fft, hop, str(g[name + "_win"]), zp)
    assert S.shape == g[name + "_S"].shape                      # integer frame/bin arithmetic exact
    assert relerr(S, g[name + "_S"]) < TOL
    m = par.fourier.get_mag(x, n_fft, hop, str(g[name + "_win"]), zp)
    assert relerr(m, np.abs(g[name + "_S"].astype(np.complex128)) + 1e-7) < TOL


def test_stft_kat_and_strided(par, golden):
    x = inputs.noise(4096, 0)
    S = par.fourier.stft(x, 1024, 256, "hann", 1)
    assert abs(S[0, 0] - (-0.19117718935012817)) < 1e-5 and abs(S[100, 8] - (-0.9019840955734253 - 0.008981410413980484j)) < 1e-5
    m = par.fourier.get_mag(x, 1024, 256, "blackmanharris", 1)
    assert relerr(m, golden["stft"]["kat2_mag"]) < TOL
    st = np.stack((inputs.noise(3000, 10), inputs.noise(3000, 11)), axis=-1)
    assert relerr(par.fourier.stft(st[:, 1], 512, 128, "blackmanharris", 1), golden["stft"]["strided_S"]) < TOL
    # device-resident channel view: element stride 2, no host copy
    t = par.torch.from_numpy(st).cuda()
    Sd = par.fourier.stft(t[:, 1], 512, 128, "blackmanharris", 1)
    assert relerr(Sd.cpu().numpy(), golden["stft"]["strided_S"]) < TOL
    with pytest.raises(ValueError):
        par.fourier.stft(st, 512, 128)
    from pyaudiorestoration_amd import _lib
    with pytest.raises(_lib.ParUnsupported):                    # reference chain would fall through here
        par.fourier.stft(x, 1000, 250)


def test_stft_vs_c_oracle_larger(par):
    from oracle import oracle_c as C
    x = inputs.pilot(200000, 192000)
    for n_fft, hop, wname in ((1024, 256, "blackmanharris"), (512, 32, "hann"), (4096, 512, "hann"), (8192, 2048, "hamming")):
        win = scipy.signal.get_window(wname, n_fft).astype(np.float32)
        ref = C.stft(x, n_fft, hop, win, 1, mode=1)
        got = par.fourier.get_mag(x, n_fft, hop, wname)
        assert got.shape == ref.shape and relerr(got, ref) < TOL, (n_fft, hop)


def test_stft_short_and_edge_inputs(par):
    from oracle import oracle_np as O
    for n, n_fft, hop in ((40, 64, 16), (33, 64, 16), (700, 1024, 256), (1024, 1024, 1024)):
        x = inputs.noise(n, 77)
        assert relerr(par.fourier.stft(x, n_fft, hop, "hann"), O.stft(x, n_fft, hop, "hann")) < TOL, (n, n_fft)


# ----------------------------------------------------------------------------------- ISTFT
@pytest.mark.parametrize("name", ["rt512", "rt1024", "rt256"])
def test_istft_golden(par, golden, name):
    g = golden["istft"]
    n, seed, n_fft, hop = (int(v) for v in g[name + "_cfg"])
    x = inputs.noise(n, seed)
    S = par.fourier.stft(x, n_fft, hop)
    S2 = S.copy()
    S2[5:40, 3:9] *= 0.25
    keep = S.copy()
    y = par.fourier.istft(S, hop_length=hop, length=n)
    assert np.array_equal(S, keep)                              # not mutated (quirk 9 decision)
    assert relerr(y, g[name + "_y"]) < TOL and relerr(y, x) < TOL
    assert relerr(par.fourier.istft(S2, hop_length=hop, length=n), g[name + "_ymod"]) < TOL
    assert relerr(par.fourier.istft(S, hop_length=hop), g[name + "_ynolen"]) < TOL


def test_istft_heal_framing_and_device_roundtrip(par, golden):
    g = golden["istft"]
    x = inputs.noise(5000, 15)
    S = par.fourier.stft(par.fourier.fix_length(x, len(x) + 256), 512, 32)
    assert tuple(g["heal_shape"]) == S.shape
    assert relerr(par.fourier.istft(S, hop_length=32, length=len(x)), g["heal_y"]) < TOL
    xt = par.torch.from_numpy(inputs.noise(100000, 16)).cuda()
    St = par.fourier.stft(xt, 1024, 256)                        # stays in HBM
    yt = par.fourier.istft(St, hop_length=256, length=xt.numel())
    assert relerr(yt.cpu().numpy(), xt.cpu().numpy()) < TOL


@pytest.mark.parametrize("n_fft,hop", [(16, 4), (64, 7), (256, 64), (512, 1), (512, 33), (1024, 256), (2048, 512),
                                       (4096, 1024), (8192, 2048), (256, 300), (128, 5000)])
def test_istft_size_matrix_vs_oracle(par, n_fft, hop):
    """Every transform size, even/odd/tiny/oversized hops: the fused LDS overlap-add (and, for (128, 5000) and 8192/2048, the
    two-kernel path whose overlap-add span does not fit LDS) against the oracle on a modified spectrogram."""
    from oracle import oracle_np as O
    n = max(6 * n_fft, 3000) + 37
    if hop == 5000:
        n = 60000
    x = inputs.noise(n, n_fft + hop)
    S = O.stft(x, n_fft, hop, "blackmanharris").astype(np.complex64)
    S[1:n_fft // 4, ::2] *= 0.5
    for length in (n, None, n // 2 + 3):
        want = O.istft(S, hop, "blackmanharris", length)
        got = par.fourier.istft(S, hop_length=hop, length=length)
        assert got.shape == want.shape and relerr(got, want) < TOL, (n_fft, hop, length)
    from pyaudiorestoration_amd import _lib
    L = _lib.lib()
    # the frame array is needed where the overlap-add span exceeds 64 KB of LDS or a frame fills the workgroup alone
    assert (L.par_istft_scratch_floats(10, n_fft, hop) != 0) == (hop == 5000 or n_fft >= 4096)


@pytest.mark.parametrize("n_fft,hop", [(16384, 4096), (65536, 16384), (1048576, 262144), (32768, 3000)])
def test_istft_above_8192_points(par, n_fft, hop):
    """r04: frames of the four-step size class (the GUI's FFT sizes go to 2^20, util/widgets.py:334-351): `istft` accepts what
    `stft` produces.  irfft through big_fft_c2c + the frame array + the gather overlap-add, against the oracle's istft on a
    modified spectrogram, and the stft -> istft round trip on the device."""
    from oracle import oracle_np as O
    n = 3 * n_fft + 12345
    x = inputs.noise(n, 7 + (n_fft >> 14))
    S = O.stft(x, n_fft, hop, "blackmanharris").astype(np.complex64)
    S[1:n_fft // 8, ::2] *= 0.5
    for length in (n, None):
        want = O.istft(S, hop, "blackmanharris", length)
        got = par.fourier.istft(S, hop_length=hop, length=length)
        assert got.shape == want.shape and relerr(got, want) < TOL, (n_fft, hop, length, relerr(got, want))
    xt = par.torch.from_numpy(x).cuda()
    St = par.fourier.stft(xt, n_fft, hop)                         # four-step forward transform, stays in HBM
    yt = par.fourier.istft(St, hop_length=hop, length=n)
    assert relerr(yt.cpu().numpy(), x) < TOL


def test_istft_single_frame_without_length_is_empty(par):
    """One frame and no explicit length: the reference trims n_fft/2 from both ends of an n_fft-long overlap-add and
    returns an empty array (found by tools/fuzz_stft.py: the device path used to reject the empty output buffer)."""
    S = np.ones((129, 1), dtype=np.complex64)
    y = par.fourier.istft(S, hop_length=64)
    assert y.shape == (0,)
    assert par.fourier.istft(S, hop_length=64, length=100).shape == (100,)


# ------------------------------------------------------------------------------- positions
def test_speed_to_pos_golden_bit_exact(par, golden):
    g = golden["speed_to_pos"]
    n = 8192
    st = np.linspace(0, n, 33)
    sp = 1 + 0.01 * np.sin(2 * np.pi * np.arange(33) / 16 + 0.7)
    pos = par.resampling.speed_to_pos(st, sp, n)
    assert len(pos) == 8191 and np.array_equal(pos, g["kat3_pos"])          # KAT3, float64 bit-exact
    assert np.array_equal(par.resampling.speed_to_pos(np.array((0.0, 20000.0)), np.array((0.5, 2.0)), 20000), g["ramp_pos"])
    sc = inputs.bench_speed_curve(2.0, 48000)
    assert np.array_equal(par.resampling.speed_to_pos(sc[:, 0] * 48000, sc[:, 1], 96000), g["bench_pos"])
    assert np.array_equal(par.resampling.speed_to_pos(g["wobble_st"], g["wobble_sp"], 30000), g["wobble_pos"])
    assert np.array_equal(par.resampling.speed_to_pos(g["untrimmed_st"], g["untrimmed_sp"], 10000), g["untrimmed_pos"])


def test_gentle_ramps_reciprocals_from_the_previous_block(par):
    """k_seg_sum starts the divisions of a gentle ramp from the reciprocals eight steps back (one Newton step + the residual
    correction instead of v_rcp_f64 + two): the positions must stay bit-identical to numpy's divisions -- speeds from 0.07 to
    50 (segments of 18 .. 12 800 steps), ramps right at the 1e-5 gate, constant segments, and a wave that mixes gentle and
    steep segments (which must take the ordinary path), through the position array and through the fused checkpoints."""
    from oracle import oracle_c as C
    import torch
    rng = np.random.default_rng(61)
    for scale, m, hop in ((1.0, 6000, 256), (0.07, 3000, 256), (50.0, 800, 256), (0.9, 5000, 64), (3.3, 2000, 1024)):
        st = np.arange(m) * float(hop)
        gentle = scale * (1.0 + 0.01 * np.sin(np.arange(m) * 0.013 + 0.4) + 1e-7 * rng.standard_normal(m))
        at_gate = scale * (1.0 + np.cumsum(rng.choice([-1.0, 0.0, 1.0], m)) * 1.2e-6 * hop / 8)       # |d8| ~ 1.2e-5 speed: both sides
        mixed = gentle.copy()
        mixed[::97] *= 1.03                                                   # one steep segment in most waves
        for sp in (gentle, at_gate, mixed):
            sp = np.maximum(sp, 0.065 * max(scale, 1.0) if scale >= 1 else 0.065)
            n_in = int(st[-1] * float(np.mean(sp)) * 0.97)
            ref, _ = C.speed_to_pos(st, sp, n_in)
            pos = par.resampling.speed_to_pos(st, sp, n_in)
            assert len(pos) == len(ref) and np.array_equal(pos, ref), (scale, hop)
            plan = par.resampling.speed_plan_dev(torch.from_numpy(st).cuda(), torch.from_numpy(sp).cuda(), n_in, fused=True,
                                                 eager=True)
            if plan.fused_ok:
                assert not plan.lazy
                buf = torch.empty(plan.len_out, dtype=torch.float64, device="cuda")
                from pyaudiorestoration_amd import _dev, _lib
                _lib.check(_lib.lib().par_speed_to_pos_fill_fused(0, _dev.ptr(plan.speeds_t), plan.m, _dev.ptr(plan.work),
                                                                  _dev.ptr(plan.aux), plan.max_out, _dev.ptr(buf), plan.len_out,
                                                                  _dev.stream_ptr(0)))
                assert np.array_equal(buf.cpu().numpy(), ref), (scale, hop, "checkpoints")


def test_speed_to_pos_vs_c_oracle_long(par):
    from oracle import oracle_c as C
    sr, dur = 192000, 20.0
    sc = inputs.bench_speed_curve(dur, sr)
    n = int(sr * dur)
    ref, trimmed = C.speed_to_pos(sc[:, 0] * sr, sc[:, 1], n)
    pos = par.resampling.speed_to_pos(sc[:, 0] * sr, sc[:, 1], n)
    assert len(pos) == len(ref) and np.array_equal(pos, ref)
    assert np.all(np.diff(pos) > 0)                                           # property: monotone
    from pyaudiorestoration_amd import _lib
    with pytest.raises(_lib.ParError):
        par.resampling.speed_to_pos(np.array([0.0, 3.0, 6.0]), np.array([0.2, 0.2, 0.2]), 100)    # n_i < 2


# ------------------------------------------------------------------------------------ sinc
def test_sinc_golden(par, golden):
    g = golden["sinc"]
    gp = golden["speed_to_pos"]
    y = par.resampling.sinc_wrapper(gp["kat3_pos"], inputs.sine(8192, 440, 44100), 0, 32)
    assert relerr(y, g["kat4_y"]) < TOL
    assert abs(y[4000] - (-0.45488033)) < 1e-5 and abs(y[100] - 0.004950763) < 1e-5      # KAT4
    assert np.allclose(y[:3], [0.91083026, 0.903764, 0.91026634], atol=1e-5)              # edge-quirk region
    sig = inputs.noise(600, 30)
    y = par.resampling.sinc_wrapper(np.arange(600, dtype=np.float64), sig, 0, 8)
    assert relerr(y, g["ident_y"]) < TOL and np.all(np.abs(y[8:] - sig[8:]) < 1e-5)      # quirk 1 kept
    sig = (inputs.sine(20000, 440, 44100, 0.5) + inputs.sine(20000, 21000, 44100, 0.1)).astype(np.float32)
    assert relerr(par.resampling.sinc_wrapper(gp["ramp_pos"], sig, 0, 50), g["ramp_y"]) < TOL
    bsig = inputs.bench_signal(0, 96000, 48000)
    yb = par.resampling.sinc_wrapper(gp["bench_pos"], bsig, 0, 32)
    assert relerr(yb, g["bench_y"]) < TOL
    assert block_relerr(yb, g["bench_y"]) < 2 * TOL            # per 4096-sample block on its own scale
    assert relerr(par.resampling.sinc_wrapper(g["tail_pos"], inputs.noise(1000, 31), 0, 16), g["tail_y"]) < TOL
    sig = inputs.noise(3000, 32)
    pos = np.cumsum(np.full(2500, 1.013)) - 0.4
    assert relerr(par.resampling.sinc_wrapper(pos, sig, 0, 1), g["nt1_y"]) < TOL
    assert relerr(par.resampling.sinc_wrapper(pos, sig, 0, 100), g["nt100_y"]) < TOL
    assert relerr(par.resampling.sinc_wrapper(g["down_pos"], sig, 0, 24), g["down_y"]) < TOL


def test_sinc_wrapper_mt_and_core_signatures(par, golden):
    g = golden["sinc"]
    gp = golden["speed_to_pos"]
    bsig = inputs.bench_signal(0, 96000, 48000)
    buf = np.zeros((len(gp["bench_pos"]), 2), dtype=np.float32)
    assert par.resampling.sinc_wrapper_mt(buf[:, 1], gp["bench_pos"], bsig, 0, 32) is None
    assert relerr(buf[:, 1], g["bench_y"]) < TOL and not buf[:, 0].any()
    out = np.empty(len(gp["bench_pos"]), dtype=np.float32)
    N = np.arange(-32, 33, dtype="float32")
    par.resampling.sinc_core(gp["bench_pos"], bsig, 0, out, np.hanning(65).astype("float32"), N)
    assert relerr(out, g["bench_y"]) < TOL
    with pytest.raises(UnboundLocalError):
        par.resampling.sinc_wrapper(np.array([5.0]), bsig, 0, 32)
    empty = par.resampling.sinc_wrapper(np.empty(0), bsig, 0, 32)       # the reference's loop never runs
    assert empty.shape == (0,) and empty.dtype == np.float32


@pytest.mark.parametrize("NT", [4, 32, 50])
def test_sinc_vs_c_oracle_regimes(par, NT):
    """fc == 1, fc < 1, mixed waves, big speeds (LDS span overflow -> global path), all vs the C oracle."""
    from oracle import oracle_c as C
    rng = np.random.default_rng(5)
    sig = inputs.bench_signal(0, 400000, 96000)
    n_out = 150000
    for name, speed in (("slow", np.full(n_out, 0.97)), ("fast", np.full(n_out, 1.031)),
                        ("mixed", 1 + 0.02 * np.sin(np.arange(n_out) * 2e-3)),
                        ("jitter", 1 + 0.3 * rng.random(n_out)), ("x2.5", np.full(n_out, 2.5)),
                        ("x9", np.full(40000, 9.0))):
        pos = 40.25 + np.cumsum(speed)
        ref = C.sinc(pos, sig, NT, threads=8)
        got = par.resampling.sinc_wrapper(pos, sig, 0, NT)
        assert relerr(got, ref) < TOL, (name, NT, relerr(got, ref))
        assert block_relerr(got, ref) < 2 * TOL, (name, NT, block_relerr(got, ref))
    # mixed-level material: a passage 60 dB below the rest is judged on its own scale
    lvl = np.where((np.arange(len(sig)) // 50000) % 2 == 0, 1.0, 1e-3).astype(np.float32)
    pos = 40.25 + np.cumsum(1 + 0.01 * np.sin(np.arange(n_out) * 2e-3))
    ref = C.sinc(pos, sig * lvl, NT, threads=8)
    got = par.resampling.sinc_wrapper(pos, sig * lvl, 0, NT)
    assert block_relerr(got, ref) < 2 * TOL, ("levels", NT, block_relerr(got, ref))


def test_sinc_exact_integer_positions_and_repeats(par):
    from oracle import oracle_c as C
    sig = inputs.noise(5000, 9)
    pos = np.concatenate((np.arange(100, 2100, dtype=np.float64), [2100.5, 2100.5, 2100.5, 2101.0],
                          np.arange(2101.0, 2600.0, 0.5)))
    assert relerr(par.resampling.sinc_wrapper(pos, sig, 0, 32), C.sinc(pos, sig, 32)) < TOL


def test_stereo_kernel_equals_two_mono_launches(par):
    """k_sinc<fused, 2 channels>: one launch for both channels of a file (shared positions / prologue / tap weights)
    equals one mono launch per channel to float32 rounding (the lane <-> output map differs, so at speed-1 crossings
    different waves take the fc == 1 shortcut) and is bit-identical across memory layouts -- interleaved and planar, fc == 1 and fc < 1 stretches, the
    leading-edge and wide-tile float64 paths, a tail shorter than a tile; resampling.run pairs channels automatically."""
    t = par.torch
    R = par.resampling
    rng = np.random.default_rng(31)
    for n, depth, NT in ((300000, 0.01, 32), (70000, 0.4, 16), (5000, 0.0, 50), (260000, 0.05, 4)):
        m = max(3, n // 200)
        st = np.linspace(0, n, m)
        sp = 1.0 + depth * np.sin(np.arange(m) * 0.21) + 0.002 * rng.standard_normal(m)
        st_t, sp_t = t.from_numpy(st).cuda(), t.from_numpy(sp).cuda()
        inter = t.from_numpy(rng.standard_normal((n, 2)).astype(np.float32)).cuda()
        plan = R.speed_plan_dev(st_t, sp_t, n, fused=True)
        assert plan.fused_ok
        L = plan.len_out
        want = [R.varispeed_fused_dev(plan, inter.reshape(-1)[c:], NT, sig_stride=2, len_in=n).clone() for c in (0, 1)]
        out = t.empty((L, 2), dtype=t.float32, device="cuda")
        R.varispeed_fused_stereo_dev(plan, inter.reshape(-1)[0:], inter.reshape(-1)[1:], NT, out.reshape(-1)[0:],
                                     out.reshape(-1)[1:], sig_stride=2, len_in=n, out_stride=2)
        for c in (0, 1):
            assert relerr(out[:, c].cpu().numpy(), want[c].cpu().numpy()) < 2e-6, (n, depth, NT, c)
        planar = inter.T.contiguous()                                      # (2, n): stride 1, separate rows
        o0, o1 = t.empty(L, dtype=t.float32, device="cuda"), t.empty(L, dtype=t.float32, device="cuda")
        R.varispeed_fused_stereo_dev(plan, planar[0], planar[1], NT, o0, o1)
        assert t.equal(o0, out[:, 0]) and t.equal(o1, out[:, 1])


def test_strong_slowdown_and_wide_tiles_found_by_fuzz(par):
    """tools/fuzz_resampler.py findings: (1) with the read head moving > 8 input samples per output (fc < 1/8) the
    output is a long average, small against the signal; float32 taps would exceed 1e-5 of the OUTPUT peak, so those
    lanes run in float64; (2) both kernel forms share one staging limit, so a tile too wide for LDS takes the float64
    path in both and the fused output stays bit-identical to the position-array output."""
    from oracle import oracle_c as C
    t = par.torch
    rng = np.random.default_rng(516428)
    n = 3000
    sig = (30 * rng.standard_normal(n)).astype(np.float32)
    st = np.linspace(0, n, 3)
    for speeds in ([0.002, 1.0, 0.002], [0.02, 0.021, 0.02], [0.11, 0.12, 0.13]):
        sp = np.array(speeds)
        pos, _ = C.speed_to_pos(st, sp, n)
        for NT in (32, 100):
            want = C.sinc(pos, sig, NT)
            got = par.resampling.sinc_wrapper(pos, sig, 0, NT)
            assert np.max(np.abs(got - want)) < 1e-5 * np.max(np.abs(want)), (speeds, NT)      # relative to the OUTPUT peak
    n = 60000
    m = 235
    st = np.linspace(0, n, m)
    sp = 0.2 * (1.0 + 0.002 * np.sin(np.arange(m) * 0.1))                 # 5 input samples per output: tiles ~5.2 k wide
    sig = rng.standard_normal(n).astype(np.float32)
    st_t, sp_t, sig_t = t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), t.from_numpy(sig).cuda()
    plan = par.resampling.speed_plan_dev(st_t, sp_t, n, fused=True)
    assert plan.fused_ok
    pos_t = par.resampling.speed_to_pos_dev(st_t, sp_t, n)
    for NT in (2, 32):
        fused = par.resampling.varispeed_fused_dev(plan, sig_t, NT)
        assert same_resample(fused, par.resampling.sinc_resample_dev(pos_t, sig_t, NT))
        assert relerr(fused.cpu().numpy(), C.sinc(pos_t.cpu().numpy(), sig, NT)) < TOL


def test_operator_slot_accepts_any_finite_positions(par):
    """The open operator slot takes ANY float64 sample_at (found by tools/fuzz_operator_slot.py): positions beyond the
    int64 range select an empty slice in the reference (Python integers) -> 0.0, never an out-of-range read;
    non-monotonic / repeated / off-the-ends arrays match the oracle; non-finite ones raise like int(round(p));
    Linear mode is bit-exact np.interp (no FMA contraction), NaN abscissae pass through."""
    from oracle import oracle_c as C
    from oracle import oracle_np as O
    t = par.torch
    rng = np.random.default_rng(4758)
    sig = rng.standard_normal(60000).astype(np.float32)
    for wild in (1e19, -1e19, 1e30, 1e300, -1e300):
        pos = np.array([100.3, 101.2, wild, 103.0, 104.9])
        got = par.resampling.sinc_wrapper(pos, sig, 0, 16)
        assert got[2] == 0.0 and np.all(np.isfinite(got))
        assert relerr(got, C.sinc(pos, sig, 16)) < TOL
    pos = rng.uniform(-60, len(sig) + 60, 40000)
    pos[::7] = pos[1::7][:len(pos[::7])]                                  # repeats -> zero periods
    assert relerr(par.resampling.sinc_wrapper(pos, sig, 0, 32), C.sinc(pos, sig, 32)) < TOL
    with pytest.raises(OverflowError):
        par.resampling.sinc_wrapper(np.array([1.0, np.inf, 3.0]), sig, 0, 8)
    with pytest.raises(ValueError):
        par.resampling.sinc_wrapper(np.array([1.0, np.nan, 3.0]), sig, 0, 8)
    lin_pos = rng.uniform(0, len(sig), 400000)
    lin_pos[:5] = [-3.0, 1e300, float(len(sig) - 1), 0.0, 59998.999999999]
    got = par.resampling.linear_resample_dev(t.from_numpy(lin_pos).cuda(), t.from_numpy(sig).cuda()).cpu().numpy()
    assert np.array_equal(got, O.linear_resample(lin_pos, sig))
    nan_out = par.resampling.linear_resample_dev(t.tensor([1.5, float("nan")], dtype=t.float64, device="cuda"),
                                                 t.from_numpy(sig).cuda()).cpu().numpy()
    assert np.isnan(nan_out[1]) and not np.isnan(nan_out[0])


def test_linear_and_lag_golden(par, golden):
    g = golden["linear_lag"]
    sig = inputs.noise(5000, 50)
    t = par.torch
    got = par.resampling.linear_resample_dev(t.from_numpy(g["pos"]).cuda(), t.from_numpy(sig).cuda()).cpu().numpy()
    assert relerr(got, g["lin"]) < 1e-6


def test_lag_curve_positions_bit_exact(par, golden, tmp_path):
    """K_lag: np.interp over arange + find_cutoff + clip restated on the device, bit for bit; and the lag branch
    of resampling.run end to end."""
    from oracle import oracle_np as O
    from pyaudiorestoration_amd import io_ops
    g = golden["linear_lag"]
    sig = inputs.noise(5000, 50)
    pos = par.resampling.lag_to_pos_dev(g["lag"], int(g["sr"]), len(sig)).cpu().numpy()
    assert pos.shape == g["pos"].shape and np.array_equal(pos, g["pos"])
    rng = np.random.default_rng(11)
    for n, m in ((100000, 40), (7777, 2), (300000, 500)):
        t = np.sort(rng.uniform(0, n / 48000, m))
        t[0] = 0.0
        lag = np.stack((t, np.cumsum(rng.normal(0, 2e-4, m))), axis=-1)
        want = O.lag_to_positions(lag, 48000, n)
        got = par.resampling.lag_to_pos_dev(lag, 48000, n).cpu().numpy()
        assert got.shape == want.shape and np.array_equal(got, want), (n, m)
    fn = str(tmp_path / "sync.wav")
    par.resampling.run((fn,), signal_data=((sig[:, None], int(g["sr"])),), lag_curve=g["lag"], resampling_mode="Linear")
    y, _, _ = io_ops.read_file(str(tmp_path / "sync_res.wav"))
    assert relerr(y[:, 0], g["lin"]) < 1e-6


def test_run_writes_res_wav_like_reference(par, golden, tmp_path):
    """resampling.run drop-in: progress callbacks, channel filtering, output naming, FLOAT wav."""
    from pyaudiorestoration_amd import io_ops
    g, gp = golden["sinc"], golden["speed_to_pos"]
    sr = 48000
    sig = np.stack((inputs.bench_signal(0, 96000, sr), inputs.noise(96000, 1)), axis=-1)
    curve = inputs.bench_speed_curve(2.0, sr)
    seen = []

    class Sig:
        class notifyProgress:
            emit = staticmethod(seen.append)
    fn = str(tmp_path / "tape.flac")
    par.resampling.run((fn,), signal_data=((sig, sr),), speed_curve=curve, resampling_mode="Sinc", sinc_quality=32,
                       use_channels=(0, 5), prog_sig=Sig, suffix="_x")
    y, sr2, ch = io_ops.read_file(str(tmp_path / "tape_res_x.wav"))
    assert sr2 == sr and ch == 1 and y.shape == (len(gp["bench_pos"]), 1)
    assert relerr(y[:, 0], g["bench_y"]) < TOL
    assert seen[0] == 0 and seen[-1] == 100 and 100.0 in seen
    par.resampling.run((fn,), signal_data=((sig, sr),), speed_curve=curve, resampling_mode="Linear")
    y, _, ch = io_ops.read_file(str(tmp_path / "tape_res.wav"))
    assert ch == 2 and relerr(y[:, 0], np.interp(gp["bench_pos"], np.arange(96000), sig[:, 0], left=0, right=0)) < 1e-6


# --------------------------------------------------------------------------------- filters
def test_filters_golden(par, golden):
    g = golden["filters"]
    x = inputs.noise(4096, 0).astype(np.float64)
    f = par.filters.butter_bandpass_filter
    assert relerr(f(x, 1000, 4000, 44100, order=3), g["band"]) < 1e-9
    assert relerr(f(x, 0, 20, 172.265625, order=3), g["low"]) < 1e-9
    assert relerr(f(x, 300, 0, 44100, order=3), g["high"]) < 1e-9
    assert relerr(f(x, 500, 2000, 44100), g["band5"]) < 1e-9
    assert f(x, 0, 0, 44100) is x
    with pytest.raises(ValueError):
        f(x[:10], 0, 20, 172.0, order=3)
    big = inputs.noise(1350000, 3).astype(np.float64)           # C2-sized speed curve
    assert relerr(f(big, 0, 20, 375.0, order=3), scipy.signal.sosfiltfilt(scipy.signal.butter(3, 20 / 187.5, btype="low", output="sos"), big)) < 1e-9
    # a whole-file filter (dropouts_gui.py:314-321 band-passes entire recordings): from 2e7 samples on the block kernels
    # go through LDS tiles and the block boundaries through the two-level chain
    huge = np.random.default_rng(9).standard_normal((1 << 25) + 777)
    want = scipy.signal.sosfiltfilt(scipy.signal.butter(3, [300 / 24000, 6000 / 24000], btype="band", output="sos"), huge)
    assert relerr(f(huge, 300, 6000, 48000, order=3), want) < 1e-9


# -------------------------------------------------------------------------------- trackers
def test_trackers_golden(par, golden):
    g = golden["trackers"]
    sr, n, n_fft, hop = (int(v) for v in g["cfg"])
    x = inputs.pilot(n, sr)
    spec = par.fourier.get_mag(x, n_fft, hop, "blackmanharris", 1)
    trail = [(0.2, 4000.0), (1.3, 4000.0)]
    for name, key, tol in (("Peak", "peak", 1e-6), ("Peak Track", "peak_track", 1e-6), ("Center of Gravity", "center_of_gravity", 1e-6),
                           ("Correlation", "correlation", 1e-5), ("Freehand Draw", "freehand_draw", 1e-12),
                           ("Zero-Crossing", "zero_crossing", 1e-6)):
        tr = par.wow.wow_detectors[name](spec, x[:, None], list(trail), n_fft, hop, sr, 0.5, "Linear")
        assert np.array_equal(tr.times, g[key + "_times"]), name
        assert relerr(tr.freqs, g[key + "_freqs"]) < tol, (name, relerr(tr.freqs, g[key + "_freqs"]))
    tr = par.wow.wow_detectors["Peak"](spec, x[:, None], [(1.2, 4030.0), (0.1, 3980.0), (0.6, 4010.0)], n_fft, hop, sr, 2.0)
    assert np.array_equal(tr.times, g["peak2_times"]) and relerr(tr.freqs, g["peak2_freqs"]) < 1e-6
    # device-resident spectrogram (no D2H/H2D of the spectrogram)
    xt = par.torch.from_numpy(x).cuda()
    spec_t = par.fourier.get_mag(xt, n_fft, hop, "blackmanharris", 1)
    tr = par.wow.wow_detectors["Peak"](spec_t, x[:, None], list(trail), n_fft, hop, sr, 0.5)
    assert relerr(tr.freqs, g["peak_freqs"]) < 1e-6
    # r03: Peak / Peak Track with the band re-read from the signal in float64 (par_track_peak_refined_f64) -- the
    # reference's numpy backend hands its trackers float64 containers; what is left against the fixture is ITS float32
    # FFT noise (a few 1e-9), two orders below the float32-spectrogram rows above.  Channel 0 of an interleaved stereo
    # tensor as a strided view gives the same numbers.
    win_t = par.torch.from_numpy(scipy.signal.get_window("blackmanharris", n_fft).astype(np.float32)).cuda()
    inter = par.torch.stack((xt, xt.flip(0)), dim=1).contiguous()
    for view in (xt, inter.reshape(-1)[0::2]):
        refine = {"x": view, "n_fft": n_fft, "zeropad": 1, "window": win_t}
        for name, key in (("Peak", "peak"), ("Peak Track", "peak_track")):
            tr = par.wow.wow_detectors[name](spec_t, x[:, None], list(trail), n_fft, hop, sr, 0.5, "Linear", refine=refine)
            assert np.array_equal(tr.times, g[key + "_times"]), name
            assert relerr(tr.freqs, g[key + "_freqs"]) < 2e-8, (name, relerr(tr.freqs, g[key + "_freqs"]))


def test_correlation_tracker_wide_and_clipped_bands(par):
    """CorrelationTracker bands wider than 682 bins used to exceed the default 64 KB of dynamic LDS at launch,
    bands over 1024 bins and bands whose widened NU passes the last bin were refused.  Now: both frames in (raised) LDS
    up to 8192 grid points, from HBM beyond that, and a band clipped at the last bin keeps the reference's unclipped
    grid (util/wow_detection.py:404-408).  Checked against the oracle's restatement on the same magnitudes."""
    from oracle import oracle_np as O
    sr = 48000
    rng = np.random.default_rng(11)
    for n_fft, hop, trail in ((8192, 2048, [(0.3, 2000.0), (1.6, 12000.0)]),          # ~1700 bins: 6800 grid points, LDS
                              (16384, 4096, [(0.4, 1500.0), (1.7, 11500.0)]),         # ~3400 bins: 13 600 grid points, HBM
                              (1024, 256, [(0.1, 23990.0), (0.5, 23999.0)])):         # widened band passes the last bin
        n = 2 * sr
        x = (0.3 * np.sin(2 * np.pi * 5000.0 * np.arange(n) / sr * (1 + 0.002 * np.sin(2 * np.pi * 3.0 * np.arange(n) / sr)))
             + 0.1 * rng.standard_normal(n)).astype(np.float32)
        mag_t = par.fourier.get_mag(par.torch.from_numpy(x).cuda(), n_fft, hop, "hann", 1)
        mag = mag_t.cpu().numpy().astype(np.float64)
        want_t, want_f = O.track_correlation(mag, list(trail), n_fft, hop, sr, 0.5)
        tr = par.wow.wow_detectors["Correlation"](mag_t, x[:, None], list(trail), n_fft, hop, sr, 0.5, "Linear")
        assert np.array_equal(tr.times, want_t)
        assert relerr(tr.freqs, want_f) < 1e-9, (n_fft, relerr(tr.freqs, want_f))


def test_tracker_error_behaviour_matches_reference(par):
    """Found by tools/fuzz_trackers.py: a trail below the transform's resolution makes the reference's band slice
    [NL:NU] empty (NL < 0 after widening) and its argmax raises ValueError -- the device reports it instead of
    clamping; a trail inside one STFT frame leaves nothing to trace: Peak returns empty, Peak Track / COG index
    freqs[0] and raise IndexError."""
    from oracle import oracle_np as O
    sr, n_fft, hop = 192000, 256, 64
    x = inputs.sine(60000, 800.0, sr, 0.5)
    mag = par.fourier.get_mag(par.torch.from_numpy(x).cuda(), n_fft, hop, "blackmanharris", 1)
    low = [(0.05, 800.0), (0.25, 800.0)]                                 # 800 Hz = bin 1.07 of a 750 Hz grid
    for name in ("Peak", "Peak Track", "Center of Gravity"):
        with pytest.raises(ValueError):
            O.TRACKERS[name](mag.cpu().numpy(), list(low), n_fft, hop, sr, 0.5)
        with pytest.raises(ValueError):
            par.wow.wow_detectors[name](mag, x[:, None], list(low), n_fft, hop, sr, 0.5, "Linear")
    # the top of the spectrum: a band that ends ON the last bin with the peak there makes is_peak() read one bin past
    # the end (IndexError); a band widened PAST the last bin no longer broadcasts against its window (ValueError)
    ny = (0.5 * np.cos(np.pi * np.arange(60000))).astype(np.float32)
    mag_ny = par.fourier.get_mag(par.torch.from_numpy(ny).cuda(), n_fft, hop, "blackmanharris", 1)
    top = [(0.05, 95000.0), (0.25, 95000.0)]
    ref = O.TRACKERS["Peak"](mag_ny.cpu().numpy(), list(top), n_fft, hop, sr, 0.5)
    tr = par.wow.wow_detectors["Peak"](mag_ny, ny[:, None], list(top), n_fft, hop, sr, 0.5, "Linear")
    assert relerr(tr.freqs, ref[1]) < 1e-6
    with pytest.raises(IndexError):
        O.TRACKERS["Peak Track"](mag_ny.cpu().numpy(), list(top), n_fft, hop, sr, 0.5)
    with pytest.raises(IndexError):
        par.wow.wow_detectors["Peak Track"](mag_ny, ny[:, None], list(top), n_fft, hop, sr, 0.5, "Linear")
    past = [(0.05, 95900.0), (0.25, 95900.0)]
    for name in ("Peak", "Peak Track", "Center of Gravity"):
        with pytest.raises(ValueError):
            O.TRACKERS[name](mag_ny.cpu().numpy(), list(past), n_fft, hop, sr, 0.01)
        with pytest.raises(ValueError):
            par.wow.wow_detectors[name](mag_ny, ny[:, None], list(past), n_fft, hop, sr, 0.01, "Linear")
    one_frame = [(0.1000, 4000.0), (0.1001, 4000.0)]
    tr = par.wow.wow_detectors["Peak"](mag, x[:, None], list(one_frame), n_fft, hop, sr, 0.5, "Linear")
    assert len(tr.freqs) == 0 and len(tr.times) == 0
    for name in ("Peak Track", "Center of Gravity"):
        with pytest.raises(IndexError):
            O.TRACKERS[name](mag.cpu().numpy(), list(one_frame), n_fft, hop, sr, 0.5)
        with pytest.raises(IndexError):
            par.wow.wow_detectors[name](mag, x[:, None], list(one_frame), n_fft, hop, sr, 0.5, "Linear")


def test_pipeline_config3_flow(par, golden):
    g = golden["pipeline"]
    sr, n, n_fft, hop = (int(v) for v in g["cfg"])
    x = inputs.pilot(n, sr)
    r = par.pipeline.respeed(x, sr, [(0.05, 4000.0), (1.45, 4000.0)], n_fft, hop, 1, "Peak", 0.5, (0, 20), 32)
    assert np.array_equal(r["times"], g["track_times"]) and relerr(r["freqs"], g["track_freqs"]) < 1e-6
    assert r["speed_curve"].shape == g["curve"].shape and relerr(r["speed_curve"][:, 1], g["curve"][:, 1]) < 1e-7
    pos = r["positions"].cpu().numpy()
    assert len(pos) == len(g["pos"]) and np.max(np.abs(pos - g["pos"])) < 1e-4
    # End to end the flow amplifies (profiles/r03_p0_sensitivity.txt, DESIGN "P0"): positions are a running sum over the
    # whole file, so whatever moves the tracked frequencies by 1e-9 moves late positions by 1e-6..1e-5 samples, and where
    # that carries a position across a half-integer the reference's interpolant (window centred on round(p),
    # util/resampling.py:60-75) jumps.  With the band magnitudes taken from the signal in float64 (r03) the pilot holds
    # TOL outright (measured 2.0e-6: what is left is the fixture's own float32 FFT noise); the half-integer clause stays
    # as the statement of WHERE a difference may come from.
    y = r["output"].cpu().numpy()[:, 0]
    same = np.rint(pos) == np.rint(g["pos"])
    assert relerr(y[same], g["y"][same]) < TOL and (~same).sum() <= 4
    assert relerr(y, g["y"]) < TOL


# ------------------------------------------------------------- properties at bench-like sizes
def test_large_properties(par):
    """10 s @192 kHz: device-born signal == oracle generator; sampled outputs == oracle at random
    indices; chunk invariance (resampling a slice of the position array gives the same samples)."""
    from oracle import oracle_c as C
    import ctypes
    from pyaudiorestoration_amd import _lib, _dev
    t = par.torch
    sr, dur = 192000, 10.0
    n = int(sr * dur)
    L = _lib.lib()
    sig_t = t.empty(n, dtype=t.float32, device="cuda")
    _lib.check(L.par_synth_signal_f32(0, _dev.ptr(sig_t), 0, n, float(sr), 0x5EED, _dev.stream_ptr(0)))
    sig = sig_t.cpu().numpy()
    assert np.max(np.abs(sig - C.synth_signal(0, n, sr))) < 1e-6
    m = int(dur * sr / 256)
    st_t = t.empty(m, dtype=t.float64, device="cuda")
    sp_t = t.empty(m, dtype=t.float64, device="cuda")
    _lib.check(L.par_synth_speed_curve_f64(0, _dev.ptr(st_t), _dev.ptr(sp_t), m, dur, float(sr), 0.01, 0.55, 0.7,
                                           _dev.stream_ptr(0)))
    st, sp = st_t.cpu().numpy(), sp_t.cpu().numpy()
    rst, rsp = C.synth_curve(m, dur, sr)
    assert np.allclose(st, rst, rtol=0, atol=1e-6) and np.allclose(sp, rsp, rtol=0, atol=1e-14)
    pos_t = par.resampling.speed_to_pos_dev(st_t, sp_t, n)
    ref_pos, _ = C.speed_to_pos(st, sp, n)
    pos = pos_t.cpu().numpy()
    assert np.array_equal(pos, ref_pos)
    out = par.resampling.sinc_resample_dev(pos_t, sig_t, 32).cpu().numpy()
    idx = np.sort(np.random.default_rng(0).choice(len(pos) - 70000, 40, replace=False))
    for i in idx:                                         # oracle on 40 windows of 2000 outputs
        ref = C.sinc(pos[i:i + 2001], sig, 32)[:2000]
        assert relerr(out[i:i + 2000], ref) < TOL
    part = par.resampling.sinc_resample_dev(pos_t[123457:323457], sig_t, 32).cpu().numpy()
    assert relerr(part[:-1], out[123457:323456]) < 2e-6     # chunk invariance (tile phase changes which waves take the fc==1 path)


def test_fill_fused_from_a_lazy_plan_fails_loudly(par):
    """a caller on the ABI-102 contract ('fused_ok != 0, so par_speed_to_pos_fill_fused is allowed') used to read
    the never-written checkpoints of a LAZY plan and got garbage positions with PAR_OK.  The fill now reads the plan header on the
    device: without checkpoints every position it writes is NaN; an eager plan fills the reference's positions as before."""
    from oracle import oracle_c as C
    from pyaudiorestoration_amd import _lib, _dev
    t = par.torch
    L = _lib.lib()
    sr, n = 48000, 400_000
    curve = inputs.bench_speed_curve(n / sr, sr)
    st_t = t.from_numpy(curve[:, 0] * sr).cuda()
    sp_t = t.from_numpy(curve[:, 1].copy()).cuda()
    ref, _ = C.speed_to_pos(curve[:, 0] * sr, curve[:, 1], n)
    for eager in (False, True):
        plan = par.resampling.speed_plan_dev(st_t, sp_t, n, fused=True, eager=eager)
        assert plan.fused_ok and plan.lazy == (not eager) and plan.len_out == len(ref)
        pos = t.full((plan.len_out,), 7.0, dtype=t.float64, device="cuda")
        _lib.check(L.par_speed_to_pos_fill_fused(0, _dev.ptr(sp_t), plan.m, _dev.ptr(plan.work), _dev.ptr(plan.aux), plan.max_out,
                                                 _dev.ptr(pos), plan.len_out, _dev.stream_ptr(0)))
        if eager:
            assert t.equal(pos.cpu(), t.from_numpy(ref))
        else:
            assert bool(t.isnan(pos).all())


def redo_list(plan, cap=4096):
    """tiles the streaming kernel handed to the block kernel's list in the last fused launch on this plan (par_fused_redo_list)"""
    from pyaudiorestoration_amd import _lib, _dev
    L = _lib.lib()
    buf = (ctypes.c_int * cap)()
    cnt = ctypes.c_int(-1)
    _lib.check(L.par_fused_redo_list(0, _dev.ptr(plan.aux), plan.max_out, plan.m, buf, cap, ctypes.byref(cnt), _dev.stream_ptr(0)))
    return cnt.value, sorted(buf[i] for i in range(min(cnt.value, cap)))


def test_full_size_benchmark_workload(par):
    """The file bench.py times, at ITS size (the timed workload was never checked at its own size): 3600 s @ 192 kHz
    mono = 691.2 M samples, +-1 % / 0.55 Hz curve sampled every 256 samples, NT = 32, through the default path (lazy plan + the
    streaming kernels).  Positions to 6.9e8 (ulp 1.2e-7), 675 k tiles, 24-tile streams and the sixth-length tail streams.
      1. whole-file positions (the float64 position array from the same curve) bit-equal to the C oracle's;
      2. the fused output against the C oracle's sinc on >= 24 windows: spread over the hour, the first tile, the tiles of the
         streaming kernel's list (window-centre ties: the block kernel's), the sixth-length tail streams, the last two full
         tiles and the partial one."""
    from oracle import oracle_c as C
    from pyaudiorestoration_amd import _lib, _dev
    t = par.torch
    sr, dur = 192000, 3600.0
    n = int(sr * dur)
    L = _lib.lib()
    sig_t = t.empty(n, dtype=t.float32, device="cuda")
    _lib.check(L.par_synth_signal_f32(0, _dev.ptr(sig_t), 0, n, float(sr), 0x5EED, _dev.stream_ptr(0)))
    m = int(dur * sr / 256)
    st_t = t.empty(m, dtype=t.float64, device="cuda")
    sp_t = t.empty(m, dtype=t.float64, device="cuda")
    _lib.check(L.par_synth_speed_curve_f64(0, _dev.ptr(st_t), _dev.ptr(sp_t), m, dur, float(sr), 0.01, 0.55, 0.7,
                                           _dev.stream_ptr(0)))
    plan = par.resampling.speed_plan_dev(st_t, sp_t, n, fused=True)
    assert plan.fused_ok and plan.lazy and plan.path == 0              # what bench.py's step gets
    ref_pos, _ = C.speed_to_pos(st_t.cpu().numpy(), sp_t.cpu().numpy(), n)
    assert plan.len_out == len(ref_pos) == 691199999
    pos_t = par.resampling.speed_to_pos_dev(st_t, sp_t, n)
    assert pos_t.numel() == len(ref_pos)
    for a in range(0, len(ref_pos), 1 << 27):                          # (compared in pieces: no second 5.5 GB host copy)
        assert t.equal(pos_t[a:a + (1 << 27)].cpu(), t.from_numpy(ref_pos[a:a + (1 << 27)])), a
    del pos_t
    out = par.resampling.varispeed_fused_dev(plan, sig_t, 32)
    assert out.numel() == plan.len_out
    n_redo, tiles = redo_list(plan)
    n_tiles = plan.len_out // 1024
    assert 0 <= n_redo <= 1024, n_redo     # window-centre ties only (324 measured: |p| to 6.9e8, tie band ~3e-7 of a sample), < 0.2 % of the 675 k tiles
    assert not t.isnan(out).any()
    W = 2000
    last = len(ref_pos) - W - 1
    starts = [int(i) for i in np.linspace(0, last, 24)]
    starts += [0, 1024 - W // 2, last, (n_tiles - 2) * 1024 - W // 2, (n_tiles - 1) * 1024 - W // 2]      # first tile and its border, the last tiles
    # the launch's last round: sixth-length (4-tile) streams over the last 2048 x 24 tiles; windows over several stream borders
    tail0 = n_tiles - 2048 * 24
    starts += [(tail0 + k) * 1024 - W // 2 for k in (0, 4, 8, 1000 * 4, 6000 * 4 + 3)]
    starts += [(tail0 - 24) * 1024 - W // 2, (tail0 - 24 * 1000) * 1024 + 37]                             # ... and the long streams in front
    starts += [max(0, T * 1024 - 200) for T in tiles[:8]]                                                 # tiles of the list
    assert len(starts) >= 36 or not tiles
    worst = 0.0
    for i in starts:
        i = min(max(int(i), 0), last)
        lo = max(0, int(ref_pos[i]) - 200)
        hi = min(n, int(ref_pos[i + W]) + 200)
        sig_win = sig_t[lo:hi].cpu().numpy()
        ref = C.sinc(ref_pos[i:i + W + 1] - lo, sig_win, 32)[:W] if lo else C.sinc(ref_pos[i:i + W + 1], sig_win, 32)[:W]
        e = relerr(out[i:i + W].cpu().numpy(), ref)
        worst = max(worst, e)
        assert e < TOL, (i, e)
    print(f"benchmark workload at full size: {len(starts)} oracle windows, worst {worst:.2e}; {n_redo} tiles through the list")


def test_full_size_config2_properties(par):
    """BASELINE config 2 at FULL size (60 min @96 kHz mono = 345.6 M samples, +-1 % curve, NT = 32) through
    size-independent properties: unit speed reproduces the input (shifted by the reference's one-sample lead),
    positions bit-equal to the C oracle's over the whole file, 24 oracle windows spread over the hour,
    linearity of the resampler, NaN containment to +-NT samples."""
    from oracle import oracle_c as C
    from pyaudiorestoration_amd import _lib, _dev
    t = par.torch
    sr, dur = 96000, 3600.0
    n = int(sr * dur)
    L = _lib.lib()
    sig_t = t.empty(n, dtype=t.float32, device="cuda")
    _lib.check(L.par_synth_signal_f32(0, _dev.ptr(sig_t), 0, n, float(sr), 0x5EED, _dev.stream_ptr(0)))
    m = int(dur * sr / 256)
    st_t = t.empty(m, dtype=t.float64, device="cuda")
    sp_t = t.empty(m, dtype=t.float64, device="cuda")
    _lib.check(L.par_synth_speed_curve_f64(0, _dev.ptr(st_t), _dev.ptr(sp_t), m, dur, float(sr), 0.01, 0.55, 0.7,
                                           _dev.stream_ptr(0)))
    # 1. unit speed: positions are 1, 2, 3, ... exactly; an integer position with fc == 1 picks one sample
    one_t = t.ones(m, dtype=t.float64, device="cuda")
    plan1 = par.resampling.speed_plan_dev(st_t, one_t, n, fused=True)
    assert plan1.fused_ok
    y1 = par.resampling.varispeed_fused_dev(plan1, sig_t, 32)
    k = plan1.len_out
    assert abs(k - n) <= 2
    d = (y1[64:k - 64] - sig_t[65:k - 63]).abs().max().item()         # away from the leading-edge quirk
    assert d < 2e-6, d
    del y1, plan1, one_t
    # 2. the +-1 % curve: whole-file positions against the C oracle, bit for bit
    plan = par.resampling.speed_plan_dev(st_t, sp_t, n, fused=True)
    assert plan.fused_ok and plan.path == 0
    pos_t = par.resampling.speed_to_pos_dev(st_t, sp_t, n)
    ref_pos, _ = C.speed_to_pos(st_t.cpu().numpy(), sp_t.cpu().numpy(), n)
    assert plan.len_out == len(ref_pos) == pos_t.numel()
    assert t.equal(pos_t.cpu(), t.from_numpy(ref_pos))
    # 3. fused output == position-array output everywhere, == oracle on 24 windows across the hour
    out = par.resampling.varispeed_fused_dev(plan, sig_t, 32)
    out2 = par.resampling.sinc_resample_dev(pos_t, sig_t, 32)
    assert same_resample(out, out2)
    del out2
    for i in np.linspace(0, len(ref_pos) - 3000, 24).astype(np.int64):
        lo = max(0, int(ref_pos[i]) - 200)
        hi = min(n, int(ref_pos[i + 2000]) + 200)
        sig_win = sig_t[lo:hi].cpu().numpy()
        if lo == 0:
            ref = C.sinc(ref_pos[i:i + 2001], sig_win, 32)[:2000]
        else:                                                            # window-local coordinates: same arithmetic
            ref = C.sinc(ref_pos[i:i + 2001] - lo, sig_win, 32)[:2000]
        assert relerr(out[i:i + 2000].cpu().numpy(), ref) < TOL, i
    # 4. linearity at full size: R(a x + b y) == a R(x) + b R(y)
    y_t = t.empty(n, dtype=t.float32, device="cuda")
    _lib.check(L.par_synth_signal_f32(0, _dev.ptr(y_t), 0, n, float(sr), 0xBEEF, _dev.stream_ptr(0)))
    mix = 0.75 * sig_t - 0.5 * y_t
    r_mix = par.resampling.varispeed_fused_dev(plan, mix, 32)
    r_y = par.resampling.varispeed_fused_dev(plan, y_t, 32)
    err = (r_mix - (0.75 * out - 0.5 * r_y)).abs().max().item()
    assert err < 5e-6, err
    del mix, r_mix, r_y, y_t
    # 5. a NaN in the input poisons exactly the outputs whose reference window signal[ind-NT : ind+NT] holds it
    bad = 200_000_000
    sig_t[bad] = float("nan")
    out_n = par.resampling.varispeed_fused_dev(plan, sig_t, 32)
    ind = t.round(pos_t).to(t.int64)                          # round-half-even like Python's round()
    expect = (ind - 32 <= bad) & (bad < ind + 32)
    assert 60 <= int(expect.sum()) <= 68
    assert t.equal(t.isnan(out_n), expect)
    # elsewhere the NaN changes nothing -- bit for bit, except in the eight-tile stream around it, whose tiles the streaming
    # kernel hands to the block kernel when it meets input float16 cannot carry (same numbers to float32 rounding)
    jn = int(t.nonzero(expect)[0])
    far = t.ones_like(expect)
    far[max(0, jn - 16384):jn + 16384] = False
    assert t.equal(out_n[far], out[far])
    near = ~far & ~expect
    assert float((out_n[near] - out[near]).abs().max()) <= FUSED_TOL * float(out.abs().max())


def test_speed_plan_device_scans_vs_serial_host_chain(par, golden):
    """The device plan (128-bit fixed-point length scan + parity-translation offset scan) must agree
    bit-for-bit with the serial host evaluation, take the device path on ordinary curves, and defer
    to the host path on exact rounding ties."""
    t = par.torch
    from oracle import oracle_c as C
    rng = np.random.default_rng(11)
    cases = []
    sc = inputs.bench_speed_curve(40.0, 192000)                      # 30000 segments, offsets cross 2^8..2^22
    cases.append((sc[:, 0] * 192000, sc[:, 1], int(192000 * 40.0)))
    st = np.linspace(0, 2_000_000, 2_000_000 // 64)                  # hop 64, noisy speeds, no trim (n_in huge)
    cases.append((st, 1 + 0.05 * np.sin(np.arange(len(st)) * 0.01) + 0.003 * rng.standard_normal(len(st)), 10**9))
    st = 1000.0 + np.cumsum(rng.uniform(100, 400, 5000))             # uneven spacing, late start
    cases.append((st, rng.uniform(0.5, 2.0, 5000), int(st[-1] * 0.8)))
    for st, sp, n_in in cases:
        info_d, info_h = {}, {}
        a = par.resampling.speed_to_pos_dev(t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), n_in, info=info_d)
        b = par.resampling.speed_to_pos_dev(t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), n_in, force_host_chain=True,
                                            info=info_h)
        ref, trimmed = C.speed_to_pos(st, sp, n_in)
        assert info_d["path"] == 0 and info_h["path"] == 1
        assert info_d["trimmed"] == info_h["trimmed"] == trimmed
        assert a.numel() == b.numel() == len(ref)
        assert np.array_equal(a.cpu().numpy(), ref) and np.array_equal(b.cpu().numpy(), ref)
    # exact ties: period 256.5, speed 1 -> every other cumulative length is k + 0.5
    st = np.arange(0, 200) * 256.5
    sp = np.ones(200)
    info = {}
    a = par.resampling.speed_to_pos_dev(t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), 10**9, info=info)
    ref, _ = C.speed_to_pos(st, sp, 10**9)
    assert info["path"] == 2 and np.array_equal(a.cpu().numpy(), ref)      # host-made lengths, device everything else
    plan = par.resampling.speed_plan_dev(t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), 10**9, fused=True)
    assert plan.path == 2 and plan.fused_ok and plan.len_out == len(ref)


def test_pipelined_varispeed_equals_two_step_path(par):
    """par_varispeed_resample_f32 (fill of chunk c+1 under the sinc of chunk c, two streams) must be
    bit-identical to par_speed_to_pos_fill + par_sinc_resample_f32, for any chunk count."""
    t = par.torch
    sr, dur = 96000, 30.0
    n = int(sr * dur)
    sc = inputs.bench_speed_curve(dur, sr)
    sig_t = t.from_numpy(inputs.bench_signal(0, n, sr)).cuda()
    st_t, sp_t = t.from_numpy(sc[:, 0] * sr).cuda(), t.from_numpy(np.ascontiguousarray(sc[:, 1])).cuda()
    pos_ref = par.resampling.speed_to_pos_dev(st_t, sp_t, n)
    out_ref = par.resampling.sinc_resample_dev(pos_ref, sig_t, 32)
    for chunks in (0, 1, 3, 8, 32):
        plan = par.resampling.speed_plan_dev(st_t, sp_t, n)
        out, pos = par.resampling.varispeed_resample_dev(plan, sig_t, 32, n_chunks=chunks)
        t.cuda.synchronize()
        assert t.equal(pos, pos_ref) and t.equal(out, out_ref), chunks


def test_heal_dropouts_config4(par, golden):
    """dropout_healer data flow: device STFT -> host per-marker targets -> device gain mask -> device ISTFT."""
    from test_oracle_golden import heal_input
    g = golden["heal"]
    sr = int(g["sr"])
    x = heal_input(sr)
    y = par.pipeline.heal_dropouts(x, sr, [tuple(m) for m in g["marks"]], 512, 32)
    assert y.shape == (30000, 1) and relerr(y[:, 0], g["y"]) < TOL
    st = np.stack((x, x[::-1].copy()), axis=-1)                     # stereo, channel views stay strided on the device
    y2 = par.pipeline.heal_dropouts(st, sr, [tuple(m) for m in g["marks"]], 512, 32, channels=(0,))
    assert relerr(y2[:, 0], g["y"]) < TOL


def test_heal_batched_markers_one_launch(par):
    """Config 4 shape: the signal tiled x8 with the markers repeated per tile -- all 24 gain boxes come from ONE
    K_heal launch (atomic max == the reference's marker-by-marker clip) and must match the oracle's serial loop."""
    from oracle import oracle_np as O
    from test_oracle_golden import heal_input
    sr, tiles = 44100, 8
    x1 = heal_input(sr)
    x = np.tile(x1, tiles)
    base = [(0.2000, 500.0, 0.2110, 6000.0, 0.5), (0.4500, 800.0, 0.4680, 9000.0, 0.5), (0.2050, 1000.0, 0.2150, 3000.0, 1.0)]
    marks = [(a0 + k * len(x1) / sr, a1, b0 + k * len(x1) / sr, b1, s) for k in range(tiles) for (a0, a1, b0, b1, s) in base]
    want = O.heal_dropouts(x, sr, marks, 512, 32)
    got = par.pipeline.heal_dropouts(x, sr, marks, 512, 32)
    assert relerr(got[:, 0], want[:, 0]) < TOL
    with pytest.raises(ValueError):
        par.pipeline.heal_dropouts(x1, sr, [(0.0, 500.0, 0.01, 6000.0, 0.5)], 512, 32)      # surrounding frames < 0


def test_dropout_detector_matches_golden(par, golden):
    """Detector: device get_mag -> K_heal band volume -> scipy valley search == fixture from the reference's get_mag."""
    g = golden["detect"]
    sr, fft, hop, t0, t1, fl, fu = g["args"]
    x = par.torch.from_numpy(inputs.detect_input(int(sr))).cuda()
    m = par.fourier.get_mag(x, int(fft), int(hop), "blackmanharris", 1)
    vol, fb = par.pipeline.band_volume_db(m, int(sr), int(fft), int(hop), t0, t1, fl, fu)
    assert fb == int(g["frame_b"]) and np.abs(vol - g["vol"]).max() < 1e-4           # dB
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        found = par.pipeline.detect_dropouts(m, int(sr), int(fft), int(hop), t0, t1, fl, fu, 20, 5)
    got = np.array([(a[0], a[1], b[0], b[1]) for a, b in found])
    assert got.shape == g["found"].shape and np.abs(got - g["found"]).max() < 1e-9
    m_np = par.pipeline.band_volume_db(m.cpu().numpy(), int(sr), int(fft), int(hop), t0, t1, fl, fu)[0]   # numpy spectrogram in
    assert np.array_equal(m_np, vol)


def test_fused_varispeed_equals_position_array_path(par):
    """Fused K_sinc (positions regenerated per tile in LDS from cumsum checkpoints) must give exactly the
    output of speed_to_pos + sinc on the materialised float64 positions, for smooth, jittery, coarse and
    short-segment curves, with and without the end trim, and on the serial-host plan path; windows of every fused output
    against the C oracle (positions by the oracle's own speed_to_pos)."""
    from oracle import oracle_c as C
    t = par.torch
    rng = np.random.default_rng(3)
    cases = []
    sr, dur = 96000, 20.0
    n = int(sr * dur)
    sc = inputs.bench_speed_curve(dur, sr)
    cases.append(("bench", sc[:, 0] * sr, np.ascontiguousarray(sc[:, 1]), n, n))
    st = np.linspace(0, 600000, 600000 // 64)                                     # hop 64, noisy, untrimmed (n_in huge)
    cases.append(("hop64", st, 1 + 0.04 * np.sin(np.arange(len(st)) * 0.01) + 0.003 * rng.standard_normal(len(st)), 700000, 10**8))
    st = np.cumsum(rng.uniform(3.0, 40.0, 30000))                                 # very short, uneven segments
    cases.append(("short", st, rng.uniform(0.8, 1.25, 30000), int(st[-1]) + 100, int(st[-1] * 0.7)))
    st = np.array([0.0, 50000.0, 120000.0])                                       # two huge segments (coarse curve)
    cases.append(("coarse", st, np.array([1.1, 0.9, 1.05]), 130000, 110000))
    for name, st, sp, n_sig, n_in in cases:
        sig_t = t.from_numpy(inputs.noise(n_sig, 5)).cuda()
        st_t, sp_t = t.from_numpy(st).cuda(), t.from_numpy(sp).cuda()
        pos_ref = par.resampling.speed_to_pos_dev(st_t, sp_t, n_in)
        out_ref = par.resampling.sinc_resample_dev(pos_ref, sig_t, 32)
        for force_host in (False, True):
            plan = par.resampling.speed_plan_dev(st_t, sp_t, n_in, fused=True, force_host_chain=force_host)
            assert plan.fused_ok and plan.len_out == pos_ref.numel(), (name, force_host)
            out = par.resampling.varispeed_fused_dev(plan, sig_t, 32)
            t.cuda.synchronize()
            assert same_resample(out, out_ref), (name, force_host, float((out - out_ref).abs().max()))
            oracle_spot(out, C.speed_to_pos(st, sp, n_in)[0], sig_t, 32)
    # other qualities through the same fused path
    sig_t = t.from_numpy(inputs.noise(n, 6)).cuda()
    st_t, sp_t = t.from_numpy(sc[:, 0] * sr).cuda(), t.from_numpy(np.ascontiguousarray(sc[:, 1])).cuda()
    pos_ref = par.resampling.speed_to_pos_dev(st_t, sp_t, n)
    pos_np, _ = C.speed_to_pos(sc[:, 0] * sr, np.ascontiguousarray(sc[:, 1]), n)
    for NT in (5, 50):
        plan = par.resampling.speed_plan_dev(st_t, sp_t, n, fused=True)
        out = par.resampling.varispeed_fused_dev(plan, sig_t, NT)
        assert same_resample(out, par.resampling.sinc_resample_dev(pos_ref, sig_t, NT))
        oracle_spot(out, pos_np, sig_t, NT)


def test_config1_and_config3_on_reference_samples(par, golden):
    """BASELINE configs 1 and 3 on the reference's own demo files (data fixtures under tests/golden/):
    FLAC decode -> get_mag -> PeakTracker -> master speed curve -> positions -> fused sinc resample."""
    import os
    from pyaudiorestoration_amd import io_ops
    from test_oracle_golden import GOLD, grid
    g = golden["samples"]
    x1, sr1, _ = io_ops.read_file(os.path.join(GOLD, "flutter.flac"))
    m = par.fourier.get_mag(x1[:, 0], 1024, 256, "hann", 1)
    assert m.shape == (513, 728) and relerr(grid(m), g["c1_grid"]) < TOL
    x3, sr3, _ = io_ops.read_file(os.path.join(GOLD, "flutter_192.flac"))
    r = par.pipeline.respeed(x3, sr3, [(0.2, 4000.0), (4.0, 4000.0)], 1024, 256, 1, "Peak", 0.5, (0, 20), 32)
    assert r["spectrum"].shape == (513, 3169) and relerr(grid(r["spectrum"].cpu().numpy()), g["c3_grid"]) < TOL
    assert np.array_equal(r["times"], g["c3_track_times"]) and relerr(r["freqs"], g["c3_track_freqs"]) < 1e-6
    assert relerr(r["speed_curve"][:, 1], g["c3_curve"][:, 1]) < 1e-7
    assert r["positions"].numel() == int(g["c3_len_pos"])
    # position drift: 1.4e-9 on the curve x 8e5 samples (the fixture's own float32 FFT noise in the tracked frequencies)
    assert np.max(np.abs(r["positions"].cpu().numpy()[::1009] - g["c3_pos_grid"])) < 1e-4
    y = r["output"].cpu().numpy()[:, 0]
    # every 8th sample of the reference's WHOLE output (101 k of 811 k samples): at most 3 of them beyond 1e-5 -- 100 such samples
    # in the file (12 expected here) fail the count; and the three 3000-sample windows of the r01 fixture, whose last one sits
    # at the file's end where the drift is largest: 6 of its samples lie beyond 1e-5 (max 1.42e-5; r03 reported these as "6 of
    # 811 k"), twice that fails
    over, worst = chain_parity(y[5::8], g["c3_y_dense"], g["c3_y_peak"], "config 3, flutter_192.flac", max_over=3)
    chain_parity(y[g["c3_sel"]], g["c3_y_sel"], g["c3_y_peak"], "config 3, the three 3000-sample windows", max_over=12, q=0.999)
    print(f"config 3 end to end: {over} of {len(g['c3_y_dense'])} compared samples beyond 1e-5 of the peak, max {worst:.2e}")
    # with the reference's exact curve the positions are bit-identical and the output within tolerance
    t = par.torch
    curve = g["c3_curve"]
    plan = par.resampling.speed_plan_dev(t.from_numpy(curve[:, 0] * sr3).cuda(), t.from_numpy(np.ascontiguousarray(curve[:, 1])).cuda(),
                                         len(x3), fused=True)
    y2 = par.resampling.varispeed_fused_dev(plan, t.from_numpy(np.ascontiguousarray(x3[:, 0])).cuda(), 32).cpu().numpy()
    assert len(y2) == int(g["c3_len_pos"]) and relerr(y2[g["c3_sel"]], g["c3_y_sel"]) < TOL


def test_unverified_checkpoints_are_never_published(par):
    """a plan whose exact chunked cumsum failed its own verification goes to the serial host path, which
    re-runs the same checkpoint pass; the checkpoints must then NOT be published as valid (the fused resampler would
    regenerate positions from them) -- callers fall back to the position array, whose plan stays good."""
    from oracle import oracle_c as C
    t = par.torch
    R = par.resampling
    n = 200000
    st, sp = np.linspace(0, n, 700), 1.0 + 0.02 * np.sin(np.arange(700) * 0.07)
    st_t, sp_t = t.from_numpy(st).cuda(), t.from_numpy(sp).cuda()
    sig = inputs.noise(n, 3)
    good = R.speed_plan_dev(st_t, sp_t, n, fused=True, force_host_chain=1)
    assert good.fused_ok and good.path == 1
    bad = R.speed_plan_dev(st_t, sp_t, n, fused=True, force_host_chain=2)
    assert not bad.fused_ok and bad.path == 1 and bad.len_out == good.len_out
    pos, _ = C.speed_to_pos(st, sp, n)
    item = (st_t, sp_t, t.from_numpy(sig).cuda())
    want = C.sinc(pos, sig, 32)
    for plan in (good, bad):                      # the batch driver's per-item launch: the bad plan takes the position array
        assert relerr(R._resample_item(plan, item, 32, 0).cpu().numpy(), want) < TOL


def _block_record_stats(par, plan):
    """Flag words of a fused plan's block records, decoded on the host (layout: csrc/pos_plan.h fused_aux_view, kRec = 32)."""
    a = plan.aux.cpu().numpy()
    ck_len, tiles = plan.max_out // 8 + plan.m + 16, plan.max_out // 1024 + 4
    off = (ck_len + tiles) * 8 + plan.m * 32 + tiles * 32
    nblk = (plan.len_out + 31) // 32
    w = a[off:off + tiles * 32 * 16].view(np.uint32).reshape(-1, 4)[:nblk, 0]
    boundary = (w & 31) + 1 < 32
    return {"blocks": nblk, "boundary": int(boundary.sum()), "slow0": int(((w >> 7) & 1).sum()),
            "slow1": int((((w >> 8) & 1) & boundary).sum()), "cubic": int(((w >> 9) & 1).sum())}


def test_block_records_cover_benchmark_and_flutter_curves(par):
    """r03 block records (32 outputs, centred polynomial, second piece with its own curvature, cubic term where the ramp
    needs it): on the benchmark curve NO block leaves the model (r02 flagged 72 % of its boundary blocks slow and sent
    63 % of K_sinc's waves through the float64 redo path); a flutter curve at 44.1 kHz sits in the cubic regime and is
    still placed from records; a violent one leaves the model block by block and takes the float64 placement -- the
    output equals the position-array path's in all three."""
    t = par.torch
    cases = []
    sr, dur = 192000, 8.0
    sc = inputs.bench_speed_curve(dur, sr)
    cases.append(("bench", sc[:, 0] * sr, np.ascontiguousarray(sc[:, 1]), int(sr * dur), 32))
    n = 44100 * 12
    st = np.linspace(0, n, n // 256)
    tt = st / 44100.0
    cases.append(("flutter", st, 1 + 0.003 * np.sin(2 * np.pi * 20.0 * tt + 0.3) + 0.002 * np.sin(2 * np.pi * 3.1 * tt), n, 32))
    cases.append(("violent", st, 1 + 0.03 * np.sin(2 * np.pi * 60.0 * tt + 0.3), n, 50))
    for name, st, sp, n_in, NT in cases:
        sig_t = t.from_numpy(inputs.noise(n_in, 7)).cuda()
        st_t, sp_t = t.from_numpy(np.ascontiguousarray(st)).cuda(), t.from_numpy(np.ascontiguousarray(sp)).cuda()
        plan = par.resampling.speed_plan_dev(st_t, sp_t, n_in, fused=True)
        assert plan.fused_ok, name
        stats = _block_record_stats(par, plan)
        if name == "bench":
            assert stats["slow0"] + stats["slow1"] <= 2 and stats["cubic"] == 0, stats     # the block of the file's last output
            assert stats["boundary"] > 0.1 * stats["blocks"]
        elif name == "flutter":
            assert stats["cubic"] > 0.5 * stats["blocks"] and stats["slow0"] + stats["slow1"] <= 2, stats
        else:
            assert stats["slow0"] > 0.2 * stats["blocks"], stats
        fused = par.resampling.varispeed_fused_dev(plan, sig_t, NT)
        pos_t = par.resampling.speed_to_pos_dev(st_t, sp_t, n_in)
        via_pos = par.resampling.sinc_resample_dev(pos_t, sig_t, NT)
        assert same_resample(fused, via_pos), (name, float((fused - via_pos).abs().max() / via_pos.abs().max()))
        oracle_spot(fused, pos_t, sig_t, NT)           # (pos_t: bit-equal to the oracle's, test_speed_to_pos_* / the full-size tests)


@pytest.mark.parametrize("NT,form", [(32, -1), (32, 0), (50, 0)])
def test_unity_path_matrix_core_bank(par, NT, form, sinc_kernel):
    """r03: on the fc = 1 path of the mono NT = 32 kernel the taps n >= 5 are a Farrow bank on the matrix cores (float16 hi/lo
    split, csrc/sinc.hip unity_far_mfma); r06: NT = 50 -- the reference's default quality (util/resampling.py:162) -- has the same
    bank in the block kernel (four K slices, 13 constant fragments); NT = 32 runs through the default (streaming) kernel and
    through the block kernel (form 0).  A tape that only runs FAST (speed 1.000 .. 1.010: every wave is on that path)
    against the C oracle, for signals that suit float16 and for ones that do not and must take the literal-FMA loops:
    full-scale noise, a Nyquist tone, a 90 dB quieter passage next to a loud one, samples of 1e5 (float16 overflow), 1e-6
    (its lo part would go subnormal), and NaN / Inf samples, whose footprint in the output must be the reference's window
    (offsets -NT .. NT-1) exactly."""
    from oracle import oracle_c as C
    t = par.torch
    sinc_kernel(form)
    sr, dur = 192000, 3.0
    n = int(sr * dur)
    m = n // 256
    st = np.linspace(0, n, m)
    sp = 1.005 + 0.005 * np.sin(2 * np.pi * 0.55 * st / sr + 0.7)
    st_t, sp_t = t.from_numpy(st).cuda(), t.from_numpy(sp).cuda()
    plan = par.resampling.speed_plan_dev(st_t, sp_t, n, fused=True)
    assert plan.fused_ok
    pos, _ = C.speed_to_pos(st, sp, n)
    rng = np.random.default_rng(12)
    base = rng.uniform(-1, 1, n).astype(np.float32)
    cases = {"noise": base, "nyquist": np.cos(np.pi * np.arange(n)).astype(np.float32)}
    quiet = base.copy()
    quiet[n // 3:2 * n // 3] *= np.float32(3e-5)
    cases["loud next to quiet"] = quiet
    cases["1e5"] = base * np.float32(1e5)
    cases["1e-6"] = base * np.float32(1e-6)
    for name, sig in cases.items():
        want = C.sinc(pos, sig, NT, threads=8)
        got = par.resampling.varispeed_fused_dev(plan, t.from_numpy(sig).cuda(), NT).cpu().numpy()
        assert relerr(got, want) < TOL, (name, relerr(got, want))
        if name == "loud next to quiet":                                  # the quiet third on its own scale: float16's 11 bits
            a, b = n // 3 + 4000, 2 * n // 3 - 4000                      # would not do, hi + lo does
            ia, ib = np.searchsorted(pos, [a, b])
            assert relerr(got[ia:ib], want[ia:ib]) < TOL
        if name in ("noise", "nyquist", "loud next to quiet"):           # every 4096-sample block on its own scale
            assert block_relerr(got, want) < 2 * TOL, (name, block_relerr(got, want))
    bad = base.copy()
    bad[100_000] = np.nan
    bad[300_001] = np.inf
    bad[400_000:400_003] = -np.inf
    want = C.sinc(pos, bad, NT, threads=8)
    got = par.resampling.varispeed_fused_dev(plan, t.from_numpy(bad).cuda(), NT).cpu().numpy()
    # an infinite sample gives +-Inf or (Inf - Inf, 0 x Inf) NaN depending on the order of the sum: the FOOTPRINT of the
    # non-finite outputs is what must agree; a NaN sample alone must give exactly the reference's NaN set
    assert np.array_equal(np.isfinite(got), np.isfinite(want)) and (~np.isfinite(want)).sum() > 150
    ok = np.isfinite(want)
    assert relerr(got[ok], want[ok]) < TOL
    only_nan = base.copy()
    only_nan[123_456] = np.nan
    want = C.sinc(pos, only_nan, NT, threads=8)
    got = par.resampling.varispeed_fused_dev(plan, t.from_numpy(only_nan).cuda(), NT).cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want)) and 2 * NT - 4 <= np.isnan(want).sum() <= 2 * NT + 2


@pytest.fixture
def sinc_kernel():
    """par_debug_sinc_kernel for the duration of a test (process-wide setting, restored afterwards)."""
    from pyaudiorestoration_amd import _lib
    L = _lib.lib()
    prev = []

    def choose(form):
        prev.append(L.par_debug_sinc_kernel(form))
    yield choose
    if prev:
        L.par_debug_sinc_kernel(prev[0])


@pytest.mark.parametrize("kernel", ["streaming", "block"])
def test_streaming_kernel_and_block_kernel(par, sinc_kernel, kernel):
    """The two K_sinc kernels a mono NT = 32 file on unit strides can take (csrc/sinc2.hip, the default since r05: one wave
    streams over eight tiles, taps |n| >= 3 of BOTH regimes on the matrix cores -- fc < 1 in its moment form: fc = 1 bank + seven
    moment filters on one image -- the rest of the file through the block kernel's tile list; csrc/sinc.hip, the block kernel,
    forced through par_debug_sinc_kernel(0)).  Against the C oracle on a fast, a slow and a mixed tape, norm-wise and per
    4096-sample block; the tile list stays short; a NaN sample poisons exactly the reference's window; short and odd-length
    files work."""
    import ctypes
    from oracle import oracle_c as C
    from pyaudiorestoration_amd import _lib, _dev
    t = par.torch
    sinc_kernel(-1 if kernel == "streaming" else 0)
    L = _lib.lib()
    sr, NT = 192000, 32
    n = 700_001
    m = n // 256
    st = np.linspace(0, n, m)
    rng = np.random.default_rng(21)
    tt = np.arange(n)
    noise = rng.standard_normal(n).astype(np.float32)
    quiet = noise.copy()
    quiet[n // 3:2 * n // 3] *= np.float32(1e-3)
    signals = {"noise": noise, "nyquist": np.cos(np.pi * tt).astype(np.float32), "0.45 fs": np.cos(0.9 * np.pi * tt + 0.2).astype(np.float32),
               "loud next to quiet": quiet}
    for cname, sp in (("fast", 1.005 + 0.005 * np.sin(2 * np.pi * 4.4 * st / sr + 0.7)),
                      ("slow", 0.995 + 0.00499 * np.sin(2 * np.pi * 4.4 * st / sr + 0.7)),
                      ("mix", 1.0 + 0.01 * np.sin(2 * np.pi * 4.4 * st / sr + 0.7)), ("unit", np.ones(m))):
        plan = par.resampling.speed_plan_dev(t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), n, fused=True)
        assert plan.fused_ok
        pos, _ = C.speed_to_pos(st, sp, n)
        for name, sig in signals.items():
            want = C.sinc(pos, sig, NT, threads=8)
            got = par.resampling.varispeed_fused_dev(plan, t.from_numpy(sig).cuda(), NT).cpu().numpy()
            assert relerr(got, want) < TOL, (cname, name, relerr(got, want))
            # per 4096-sample block (stricter than the contract's norm-wise 1e-5): the block kernel's fc = 1 bank carries a LOUD
            # sample's float16 hi + lo rounding (2^-22 of it) into the window of a quiet output next to it -- 2.7e-5 of that
            # block's own peak where the level drops by 60 dB within a window (r05: first time the block kernel sees this signal)
            lim = 4 * TOL if (kernel == "block" and name == "loud next to quiet") else 2 * TOL
            assert block_relerr(got, want) < lim, (cname, name, block_relerr(got, want))
        redo = ctypes.c_int(-1)
        _lib.check(L.par_fused_redo_tiles(0, _dev.ptr(plan.aux), plan.max_out, plan.m, ctypes.byref(redo), _dev.stream_ptr(0)))
        if kernel == "streaming":
            # the odd rounding tie: the file's end tiles no longer come through the list (the launch's first workgroups do them)
            assert 0 <= redo.value <= 4, (cname, redo.value)
        else:
            assert redo.value == 0, (cname, redo.value)             # the plan zeroed the list and nobody filled it
        if cname == "mix":
            bad = noise.copy()
            bad[345_678] = np.nan
            want = C.sinc(pos, bad, NT, threads=8)
            got = par.resampling.varispeed_fused_dev(plan, t.from_numpy(bad).cuda(), NT).cpu().numpy()
            assert np.array_equal(np.isnan(got), np.isnan(want)) and 60 <= np.isnan(want).sum() <= 66
            ok = ~np.isnan(want)
            assert relerr(got[ok], want[ok]) < TOL
            loud = (noise * np.float32(2.0e4)).astype(np.float32)   # int16-scaled material: still float16's range
            want = C.sinc(pos, loud, NT, threads=8)
            got = par.resampling.varispeed_fused_dev(plan, t.from_numpy(loud).cuda(), NT).cpu().numpy()
            assert relerr(got, want) < TOL
            big = (noise * np.float32(1.0e5)).astype(np.float32)    # beyond it: every tile through the block kernel
            want = C.sinc(pos, big, NT, threads=8)
            got = par.resampling.varispeed_fused_dev(plan, t.from_numpy(big).cuda(), NT).cpu().numpy()
            assert relerr(got, want) < TOL
    # a file shorter than one wave's range, one of a few tiles, and a short INPUT stretched over many outputs (the streaming
    # kernel's ring reads whole 128-sample chunks: such files stay with the block kernel)
    for n_s, speed in ((5000, 0.997), (40_000, 0.997), (3000, 0.2)):
        m_s = max(4, n_s // 256)
        st_s = np.linspace(0, n_s, m_s)
        sp_s = speed + 0.002 * np.sin(np.arange(m_s) * 0.3)
        plan = par.resampling.speed_plan_dev(t.from_numpy(st_s).cuda(), t.from_numpy(sp_s).cuda(), n_s, fused=True)
        pos, _ = C.speed_to_pos(st_s, sp_s, n_s)
        want = C.sinc(pos, noise[:n_s], NT, threads=4)
        got = par.resampling.varispeed_fused_dev(plan, t.from_numpy(noise[:n_s].copy()).cuda(), NT).cpu().numpy()
        assert relerr(got, want) < TOL, n_s


def test_stereo_streaming_kernel_against_the_oracle(par):
    """The streaming kernel's stereo form (csrc/sinc2.hip, k_sinc_pipe<2>: interleaved NT = 32 files, the default
    since r05; one placement for both channels, which take turns in one set of bank rows; the file's end tiles by the launch's
    first workgroups).  Each channel against the C oracle on a fast, a slow, a mixed and a unit tape, norm-wise and per
    4096-sample block, with DIFFERENT material in the two channels (a channel must not leak into the other: a full-scale Nyquist
    tone beside noise, a passage 60 dB down beside a loud one); a NaN in one channel poisons exactly the reference's window of
    that channel; material beyond float16's range takes the block kernel's tile list; odd lengths and short files work."""
    import ctypes
    from oracle import oracle_c as C
    from pyaudiorestoration_amd import _lib, _dev
    t = par.torch
    R = par.resampling
    L = _lib.lib()
    sr, NT = 192000, 32
    n = 700_001
    m = n // 256
    st = np.linspace(0, n, m)
    rng = np.random.default_rng(22)
    tt = np.arange(n)
    noise = rng.standard_normal(n).astype(np.float32)
    quiet = rng.standard_normal(n).astype(np.float32)
    quiet[n // 3:2 * n // 3] *= np.float32(1e-3)
    pairs = {"noise | nyquist": (noise, np.cos(np.pi * tt).astype(np.float32)),
             "0.45 fs | loud next to quiet": (np.cos(0.9 * np.pi * tt + 0.2).astype(np.float32), quiet)}

    def run(plan, a, b):
        inter = t.from_numpy(np.stack((a, b), axis=1)).cuda().reshape(-1)
        out = t.empty((plan.len_out, 2), dtype=t.float32, device="cuda")
        R.varispeed_fused_stereo_dev(plan, inter[0:], inter[1:], NT, out.reshape(-1)[0:], out.reshape(-1)[1:], sig_stride=2,
                                     len_in=len(a), out_stride=2)
        return out.cpu().numpy()
    for cname, sp in (("fast", 1.005 + 0.005 * np.sin(2 * np.pi * 4.4 * st / sr + 0.7)),
                      ("slow", 0.995 + 0.00499 * np.sin(2 * np.pi * 4.4 * st / sr + 0.7)),
                      ("mix", 1.0 + 0.01 * np.sin(2 * np.pi * 4.4 * st / sr + 0.7)), ("unit", np.ones(m))):
        plan = R.speed_plan_dev(t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), n, fused=True)
        assert plan.fused_ok
        pos, _ = C.speed_to_pos(st, sp, n)
        for name, (a, b) in pairs.items():
            got = run(plan, a, b)
            for c, x in enumerate((a, b)):
                want = C.sinc(pos, x, NT, threads=8)
                assert relerr(got[:, c], want) < TOL, (cname, name, c, relerr(got[:, c], want))
                assert block_relerr(got[:, c], want) < 2 * TOL, (cname, name, c, block_relerr(got[:, c], want))
        redo = ctypes.c_int(-1)
        _lib.check(L.par_fused_redo_tiles(0, _dev.ptr(plan.aux), plan.max_out, plan.m, ctypes.byref(redo), _dev.stream_ptr(0)))
        assert 0 <= redo.value <= 4, (cname, redo.value)            # the odd rounding tie
        if cname == "mix":
            bad = noise.copy()
            bad[345_678] = np.nan
            got = run(plan, quiet, bad)
            want = C.sinc(pos, bad, NT, threads=8)
            assert np.array_equal(np.isnan(got[:, 1]), np.isnan(want)) and 60 <= np.isnan(want).sum() <= 66
            assert not np.isnan(got[:, 0]).any() and relerr(got[:, 0], C.sinc(pos, quiet, NT, threads=8)) < TOL
            ok = ~np.isnan(want)
            assert relerr(got[:, 1][ok], want[ok]) < TOL
            big = (noise * np.float32(1.0e5)).astype(np.float32)    # one channel beyond float16's range: every tile through the list
            got = run(plan, big, quiet)
            assert relerr(got[:, 0], C.sinc(pos, big, NT, threads=8)) < TOL and relerr(got[:, 1], C.sinc(pos, quiet, NT, threads=8)) < TOL
            _lib.check(L.par_fused_redo_tiles(0, _dev.ptr(plan.aux), plan.max_out, plan.m, ctypes.byref(redo), _dev.stream_ptr(0)))
            assert redo.value >= plan.len_out // 1024 - 4
    # a few tiles only, an odd length, a short input stretched over many outputs
    for n2, speed in ((5000, 1.003), (4 * 1024 + 77, 0.997), (9001, 0.25)):
        st2 = np.linspace(0, n2, 40)
        sp2 = np.full(40, speed) + 0.001 * np.sin(np.arange(40))
        a, b = rng.standard_normal(n2).astype(np.float32), rng.standard_normal(n2).astype(np.float32)
        plan = R.speed_plan_dev(t.from_numpy(st2).cuda(), t.from_numpy(sp2).cuda(), n2, fused=True)
        pos, _ = C.speed_to_pos(st2, sp2, n2)
        got = run(plan, a, b)
        assert got.shape[0] == len(pos)
        for c, x in enumerate((a, b)):
            assert relerr(got[:, c], C.sinc(pos, x, NT)) < TOL, (n2, speed, c)


def test_one_channel_of_an_interleaved_file_through_the_streaming_kernel(par):
    """The reference's use_channels (util/resampling.py:211-227: sinc_wrapper_mt(output[:, out_channel], sample_at, signal[:, in_channel]))
    on a two-channel file: a strided column view in, a column of the (length, n_used) output array out.  k_sinc_pipe<2, 3> (r06)
    rings the frames as they lie in memory and banks one channel.  Each channel against the C oracle on fast / slow / mixed tapes
    with different material per channel, output strides 1 (one channel used) and 2 (both used, one at a time), the OTHER channel
    holding a NaN (must not leak), an odd length; the tile diagnostic shows that the streams took the file."""
    from oracle import oracle_c as C
    from pyaudiorestoration_amd import _lib, _dev
    t = par.torch
    L = _lib.lib()
    n, sr = 777_777, 192000
    m = n // 256
    st = np.linspace(0, n, m)
    tt = st / sr
    rng = np.random.default_rng(123)
    k = np.arange(n)
    left = (rng.standard_normal(n) * np.where((k > n // 3) & (k < n // 2), 1e-3, 1.0)).astype(np.float32)
    right = np.cos(np.pi * k).astype(np.float32)                         # a full-scale Nyquist tone beside noise
    frames = np.stack([left, right], axis=1).copy()
    x = t.from_numpy(frames).cuda().reshape(-1)
    redo = ctypes.c_int(-1)
    for cname, sp in (("mixed", 1.0 + 0.01 * np.sin(2 * np.pi * 4.4 * tt + 0.7)), ("fast", 1.005 + 0.005 * np.sin(2 * np.pi * 3.0 * tt)),
                      ("slow", 0.995 + 0.0049 * np.sin(2 * np.pi * 3.0 * tt + 1.0))):
        plan = par.resampling.speed_plan_dev(t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), n, fused=True)
        pos, _ = C.speed_to_pos(st, sp, n)
        assert plan.len_out == len(pos)
        for ch, src in ((0, left), (1, right)):
            want = C.sinc(pos, src, 32, threads=8)
            for out_stride in (1, 2):
                out = t.full((plan.len_out * out_stride,), 7.0, dtype=t.float32, device="cuda")
                par.resampling.varispeed_fused_dev(plan, x[ch:], 32, out[(ch if out_stride == 2 else 0):], sig_stride=2, len_in=n,
                                                   out_stride=out_stride)
                _lib.check(L.par_fused_redo_tiles(0, _dev.ptr(plan.aux), plan.max_out, plan.m, ctypes.byref(redo), _dev.stream_ptr(0)))
                assert 0 <= redo.value <= 8, (cname, ch, redo.value)     # the streams took the file (ties only in the list)
                got = out.cpu().numpy()
                if out_stride == 2:
                    assert np.all(got[(1 - ch)::2] == 7.0)               # the other column untouched
                    got = got[ch::2]
                assert relerr(got, want) < TOL and block_relerr(got, want) < 2 * TOL, (cname, ch, out_stride, relerr(got, want))
    # a NaN in the OTHER channel stays there
    bad = frames.copy()
    bad[400_000, 1] = np.nan
    xb = t.from_numpy(bad).cuda().reshape(-1)
    out = par.resampling.varispeed_fused_dev(plan, xb[0:], 32, sig_stride=2, len_in=n).cpu().numpy()
    assert not np.isnan(out).any() and relerr(out, C.sinc(pos, left, 32, threads=8)) < TOL
    out = par.resampling.varispeed_fused_dev(plan, xb[1:], 32, sig_stride=2, len_in=n).cpu().numpy()
    ind = np.rint(pos).astype(np.int64)
    assert np.array_equal(np.isnan(out), (ind - 32 <= 400_000) & (400_000 < ind + 32))


def test_grouped_launches_equal_file_by_file(par):
    """par_varispeed_fused_batch_f32 / the batch driver's groups (r06): several planned files in one merged K_sinc launch -- mono
    files of different lengths and curves, interleaved stereo files, a mixed sequence (classes change: groups end there), a file
    too short for the streaming kernel inside a group -- give BIT-identical outputs to the ungrouped launches (each file's streams
    are cut as in its own launch), and one of them is checked against the C oracle."""
    from oracle import oracle_c as C
    t = par.torch
    R = par.resampling
    rng = np.random.default_rng(77)
    sr = 192000

    def curve(n, k):
        m = n // 256
        st = np.linspace(0, n, m)
        sp = 1.0 + 0.01 * np.sin(2 * np.pi * (2.0 + k) * st / sr + 0.7 + k)
        return t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), st, sp
    mono, stereo = [], []
    for k, n in enumerate((300_000, 777_777, 4_000, 1_234_567, 650_001, 300_000, 411_111, 902_000, 333_333)):
        st_t, sp_t, st, sp = curve(n, k)
        mono.append((st_t, sp_t, t.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()))
        if k == 1:
            keep = (st, sp, n, mono[-1][2].cpu().numpy())
    for k, n in enumerate((500_000, 250_001, 800_000, 612_345, 377_000)):
        st_t, sp_t, _, _ = curve(n, 10 + k)
        stereo.append((st_t, sp_t, t.from_numpy(rng.standard_normal((n, 2)).astype(np.float32)).cuda()))
    for items in (mono, stereo, mono[:3] + stereo[:2] + mono[3:5] + stereo[2:]):
        ref = [o.clone() for _, o, _ in R.varispeed_batch_dev(items, 32, group=1)]
        for g in (None, 2, 8):
            got = list(R.varispeed_batch_dev(items, 32, group=g))
            assert [k for k, _, _ in got] == list(range(len(items)))
            for (k, o, plan), want in zip(got, ref):
                assert o.shape == want.shape and t.equal(o, want), (g, k)
    st, sp, n, sig = keep
    pos, _ = C.speed_to_pos(st, sp, n)
    out1 = [o for _, o, _ in R.varispeed_batch_dev(mono, 32, group=8)][1].cpu().numpy()
    assert relerr(out1, C.sinc(pos, sig, 32, threads=8)) < TOL


def test_streaming_kernel_files_of_whole_tiles(par):
    """Files whose output is a whole number of 1024-output tiles (no partial tile: the streaming launch then has three end
    tiles instead of four), down to the smallest file the streaming kernel takes (four tiles: only tile 1 is streamed), and
    their neighbours in length -- mono and interleaved stereo against the C oracle."""
    from oracle import oracle_c as C
    t = par.torch
    R = par.resampling
    rng = np.random.default_rng(5)
    whole = 0
    for n in (4096, 4097, 5121, 8193, 8197, 20481, 20482):
        for speed in (1.0, 0.9995):
            st = np.linspace(0, n, 40)
            sp = np.full(40, speed)
            pos, _ = C.speed_to_pos(st, sp, n)
            plan = R.speed_plan_dev(t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), n, fused=True)
            assert plan.len_out == len(pos)
            whole += plan.len_out % 1024 == 0
            a, b = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
            want_a, want_b = C.sinc(pos, a, 32), C.sinc(pos, b, 32)
            assert relerr(R.varispeed_fused_dev(plan, t.from_numpy(a).cuda(), 32).cpu().numpy(), want_a) < TOL, (n, speed)
            inter = t.from_numpy(np.stack((a, b), axis=1)).cuda().reshape(-1)
            out = t.empty((plan.len_out, 2), dtype=t.float32, device="cuda")
            R.varispeed_fused_stereo_dev(plan, inter[0:], inter[1:], 32, out.reshape(-1)[0:], out.reshape(-1)[1:], sig_stride=2,
                                         len_in=n, out_stride=2)
            o = out.cpu().numpy()
            assert relerr(o[:, 0], want_a) < TOL and relerr(o[:, 1], want_b) < TOL, (n, speed)
    assert whole >= 4


def test_kernel_choice_of_the_fused_entry_point(par, sinc_kernel):
    """par_varispeed_fused_f32 picks the streaming kernel for mono NT = 32 unit-stride files (the tile diagnostic says so) and the
    block kernel for everything else; forcing the block kernel changes the mono NT = 32 result by float32 rounding only and
    nothing else at all; results against the C oracle either way."""
    import ctypes
    from oracle import oracle_c as C
    from pyaudiorestoration_amd import _lib, _dev
    t = par.torch
    L = _lib.lib()
    n, sr = 400_000, 192000
    m = n // 256
    st = np.linspace(0, n, m)
    sp = 1.0 + 0.01 * np.sin(2 * np.pi * 4.4 * st / sr + 0.7)
    sig = np.random.default_rng(77).standard_normal(n).astype(np.float32)
    plan = par.resampling.speed_plan_dev(t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), n, fused=True)
    pos, _ = C.speed_to_pos(st, sp, n)
    redo = ctypes.c_int(-1)
    st2 = np.stack([sig, sig[::-1]], axis=1).copy()
    x = t.from_numpy(st2).cuda().reshape(-1)
    res = {}
    for form in (-1, 0):
        sinc_kernel(form)
        for NT in (32, 50, 16):
            res[form, NT] = par.resampling.varispeed_fused_dev(plan, t.from_numpy(sig).cuda(), NT).cpu().numpy()
            _lib.check(L.par_fused_redo_tiles(0, _dev.ptr(plan.aux), plan.max_out, plan.m, ctypes.byref(redo), _dev.stream_ptr(0)))
            if form == -1 and NT == 32:
                assert 0 <= redo.value <= 8                  # rounding ties (the file's end tiles are done inside the streaming launch)
            elif NT == 32:
                assert redo.value == 0 or form == 0          # (the list keeps the streaming launch's count until the next plan)
        # one channel of the two-channel interleaved file (the reference's use_channels view): since r06 the streaming kernel's
        # one-channel-of-frames form at NT = 32 (the tile diagnostic says so); a channel of a THREE-channel file: the block kernel
        res[form, "strided"] = par.resampling.varispeed_fused_dev(plan, x[1:], 32, sig_stride=2, len_in=n).cpu().numpy()
        _lib.check(L.par_fused_redo_tiles(0, _dev.ptr(plan.aux), plan.max_out, plan.m, ctypes.byref(redo), _dev.stream_ptr(0)))
        assert (0 <= redo.value <= 8) if form == -1 else True
        x3 = t.from_numpy(np.stack([sig[::-1], sig, sig], axis=1).copy()).cuda().reshape(-1)
        res[form, "strided3"] = par.resampling.varispeed_fused_dev(plan, x3[1:], 32, sig_stride=3, len_in=n).cpu().numpy()
        # both channels of the interleaved file in one launch: the streaming kernel's stereo form at NT = 32, else the block kernel
        for NT in (32, 50):
            o2 = t.empty((plan.len_out, 2), dtype=t.float32, device="cuda")
            par.resampling.varispeed_fused_stereo_dev(plan, x[0:], x[1:], NT, o2.reshape(-1)[0:], o2.reshape(-1)[1:], sig_stride=2,
                                                      len_in=n, out_stride=2)
            res[form, "stereo", NT] = o2.cpu().numpy()
    for NT in (32, 50, 16):
        want = C.sinc(pos, sig, NT, threads=8)
        for form in (-1, 0):
            assert relerr(res[form, NT], want) < TOL and block_relerr(res[form, NT], want) < 2 * TOL, (form, NT)
    assert relerr(res[-1, 32], res[0, 32]) < 5e-6 and not np.array_equal(res[-1, 32], res[0, 32])
    assert np.array_equal(res[-1, 50], res[0, 50]) and np.array_equal(res[-1, 16], res[0, 16])
    assert relerr(res[-1, "strided"], res[0, "strided"]) < 5e-6 and not np.array_equal(res[-1, "strided"], res[0, "strided"])
    assert np.array_equal(res[-1, "strided3"], res[0, "strided3"])
    assert relerr(res[-1, "stereo", 32], res[0, "stereo", 32]) < 5e-6 and not np.array_equal(res[-1, "stereo", 32], res[0, "stereo", 32])
    assert np.array_equal(res[-1, "stereo", 50], res[0, "stereo", 50])
    for form in (-1, 0):
        for c in (0, 1):
            want = C.sinc(pos, st2[:, c].copy(), 32, threads=8)
            assert relerr(res[form, "stereo", 32][:, c], want) < TOL and block_relerr(res[form, "stereo", 32][:, c], want) < 2 * TOL, (form, c)
    for form in (-1, 0):
        want = C.sinc(pos, st2[:, 1].copy(), 32, threads=8)
        assert relerr(res[form, "strided"], want) < TOL and block_relerr(res[form, "strided"], want) < 2 * TOL, form
    assert relerr(res[0, "strided3"], C.sinc(pos, sig, 32, threads=8)) < TOL


def test_fused_extreme_curves_and_channels(par):
    """Fused path under stress: fast curves whose tiles overflow the LDS stage (float64 slow path),
    slow curves (many outputs per input), stereo strided views, tiny NT and NT = 100."""
    t = par.torch
    from oracle import oracle_c as C
    rng = np.random.default_rng(8)
    sig = np.stack((inputs.noise(400000, 1), inputs.noise(400000, 2)), axis=-1)
    sig_t = t.from_numpy(sig).cuda()
    for name, lo, hi, hop, NT in (("fast", 2.5, 6.0, 128, 16), ("slow", 0.3, 0.6, 256, 32), ("wide", 0.5, 2.0, 32, 100),
                                  ("nt1", 0.9, 1.1, 256, 1)):
        m = 300000 // hop
        st = np.linspace(0, 300000, m)
        sp = np.clip(0.5 * (lo + hi) + 0.5 * (hi - lo) * np.sin(np.arange(m) * 0.05) + 0.01 * rng.standard_normal(m), lo, hi)
        st_t, sp_t = t.from_numpy(st).cuda(), t.from_numpy(sp).cuda()
        n_in = 300000
        pos_ref = par.resampling.speed_to_pos_dev(st_t, sp_t, n_in)
        plan = par.resampling.speed_plan_dev(st_t, sp_t, n_in, fused=True)
        assert plan.fused_ok and plan.len_out == pos_ref.numel(), name
        out = t.zeros((plan.len_out, 2), dtype=t.float32, device="cuda")
        ref = t.zeros_like(out)
        for c in range(2):
            par.resampling.varispeed_fused_dev(plan, sig_t.reshape(-1)[c:], NT, out.reshape(-1)[c:], sig_stride=2,
                                               len_in=400000, out_stride=2)
            par.resampling.sinc_resample_dev(pos_ref, sig_t.reshape(-1)[c:], NT, ref.reshape(-1)[c:], sig_stride=2,
                                             len_in=400000, out_stride=2)
        t.cuda.synchronize()
        assert same_resample(out, ref), name
        # and the position-array path itself against the C oracle on a slice
        pos = pos_ref.cpu().numpy()
        k = min(len(pos) - 1, 60000)
        assert relerr(ref[:k, 1].cpu().numpy(), C.sinc(pos[:k + 1], sig[:, 1].copy(), NT, threads=8)[:k]) < TOL, name


def test_batch_pipeline_equals_item_by_item(par):
    """varispeed_batch_dev (plan of item k+1 on a side stream under K_sinc of item k, double-buffered plan slots that
    grow) gives bit-identical outputs to planning and resampling each item on its own, in order, including items
    produced lazily on the main stream by the iterator."""
    t = par.torch
    R = par.resampling
    rng = np.random.default_rng(8)
    specs = [(300000, 0.01), (50000, 0.3), (1200000, 0.02), (1200000, 0.02), (7000, 0.0), (640000, 0.15)]

    def make(i):
        n, depth = specs[i]
        m = max(3, n // 256)
        st = np.linspace(0, n, m)
        sp = 1.0 + depth * np.sin(np.arange(m) * 0.37 + i) + 0.001 * rng.standard_normal(m)
        sig = inputs.noise(n, 100 + i)
        return t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), t.from_numpy(sig).cuda()    # uploads on the main stream

    items = [make(i) for i in range(len(specs))]
    want = []
    for st_t, sp_t, sig_t in items:
        plan = R.speed_plan_dev(st_t, sp_t, sig_t.numel(), fused=True)
        want.append(R.varispeed_fused_dev(plan, sig_t, 16).clone())
    rng = np.random.default_rng(8)                       # same random curves for the lazy producer
    got = [(k, out.clone(), plan.len_out) for k, out, plan in R.varispeed_batch_dev((make(i) for i in range(len(specs))), 16)]
    t.cuda.synchronize()
    assert [k for k, _, _ in got] == list(range(len(specs)))
    for (k, out, n_out), w in zip(got, want):
        assert n_out == w.numel() and t.equal(out, w), k
    assert list(R.varispeed_batch_dev([], 16)) == []
    # the driver's plan buffers, slot events and planner streams live on between calls (r06): other planner counts, an abandoned
    # generator (its ring stays busy: the next call makes a private one) and a release in between change nothing
    for planners in (1, 5, 2):
        got = [out.clone() for _, out, _ in R.varispeed_batch_dev(items, 16, planners=planners)]
        assert all(t.equal(o, w) for o, w in zip(got, want)), planners
    gen = R.varispeed_batch_dev(items, 16)
    first = next(gen)[1].clone()
    got = [out.clone() for _, out, _ in R.varispeed_batch_dev(items, 16)]          # while `gen` still holds the cached ring
    assert t.equal(first, want[0]) and all(t.equal(o, w) for o, w in zip(got, want))
    gen.close()
    R.release_plan_rings()
    got = [out.clone() for _, out, _ in R.varispeed_batch_dev(items, 16)]
    assert all(t.equal(o, w) for o, w in zip(got, want))


def test_degenerate_segment_behind_the_trim_is_harmless(par):
    """A segment of fewer than 2 outputs makes the reference divide by zero -- when it gets there.  A curve that runs
    past the end of the file is trimmed first, so such a segment BEHIND the trim never exists for the reference: the
    device plan decides that itself (path 0, positions bit-identical to the C oracle); in front of the trim both refuse."""
    from oracle import oracle_c as C
    t = par.torch
    n = 100_000
    st = np.linspace(0.0, 2.0 * n, 201)
    from pyaudiorestoration_amd import _lib
    # slow = 1e-3 makes the neighbouring lengths exact rounding ties (500.5): those plans take the host-made lengths
    # (path 2), which must treat the degenerate segment the same way
    for bad_at, slow, want_path in ((150, 1.2e-3, 0), (101, 1.2e-3, 0), (150, 1e-3, 2), (50, 1.2e-3, None), (50, 1e-3, None)):
        sp = np.ones(201)
        sp[bad_at] = sp[bad_at + 1] = slow                      # 1000 samples * ~1e-3 -> n_i = 1
        st_t, sp_t = t.from_numpy(st).cuda(), t.from_numpy(sp).cuda()
        if want_path is not None:
            ref, trimmed = C.speed_to_pos(st, sp, n)
            info = {}
            pos = par.resampling.speed_to_pos_dev(st_t, sp_t, n, info=info)
            assert trimmed and info["path"] == want_path and t.equal(pos.cpu(), t.from_numpy(ref)), (bad_at, slow, info)
            plan = par.resampling.speed_plan_dev(st_t, sp_t, n, fused=True)
            assert plan.path == want_path and plan.fused_ok and plan.len_out == len(ref)
            sig_t = t.from_numpy(inputs.noise(n, 3)).cuda()
            out16 = par.resampling.varispeed_fused_dev(plan, sig_t, 16)
            assert same_resample(out16, par.resampling.sinc_resample_dev(pos, sig_t, 16))
            oracle_spot(out16, ref, sig_t, 16)
        else:
            with pytest.raises(ValueError):
                C.speed_to_pos(st, sp, n)
            with pytest.raises(_lib.ParError):
                par.resampling.speed_plan_dev(st_t, sp_t, n)


def test_host_batch_driver_equals_device_batch(par):
    """varispeed_batch_host (host files in, pinned host buffers out; upload of file k+1 and download of file k on their
    own streams) returns exactly what the device-resident batch pipeline computes -- for pageable numpy input (staged
    through the pinned ring), pinned tensors (uploaded in place), mono, stereo and 3-channel interleaved files, slots
    that grow and shrink, and it hands the outputs out in order."""
    t = par.torch
    R = par.resampling
    rng = np.random.default_rng(21)
    specs = [(300000, 1, False), (150000, 2, False), (900000, 1, True), (40000, 3, False), (900000, 2, True), (5000, 1, False)]
    host_items, dev_items = [], []
    for i, (n, ch, pin) in enumerate(specs):
        m = max(3, n // 256)
        st = np.linspace(0, n, m)
        sp = 1.0 + 0.05 * np.sin(np.arange(m) * 0.21 + i) + 0.001 * rng.standard_normal(m)
        sig = rng.standard_normal((n, ch) if ch > 1 else n).astype(np.float32)
        host_items.append((st, sp, t.from_numpy(sig).pin_memory() if pin else sig))
        dev_items.append((t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), t.from_numpy(sig).cuda()))
    want = [out.clone() for _, out, _ in R.varispeed_batch_dev(dev_items, 16)]
    # the number of planner threads changes the schedule, never a result (r05: 1 = the double-buffered pipeline of r02-r04)
    for planners in (1, 2, 5):
        got = [out.clone() for _, out, _ in R.varispeed_batch_dev(dev_items, 16, planners=planners)]
        assert len(got) == len(want) and all(t.equal(a, b) for a, b in zip(got, want)), planners
    gen = R.varispeed_batch_dev(dev_items, 16)                  # a consumer that stops early leaves no thread or plan behind
    next(gen)
    gen.close()
    got = []
    for k, out in R.varispeed_batch_host(iter(host_items), 16):
        assert out.is_pinned() and out.device.type == "cpu"
        got.append((k, out.clone()))                     # the slot is reused two files later
    assert [k for k, _ in got] == list(range(len(specs)))
    for (k, out), w in zip(got, want):
        assert out.shape == w.shape and t.equal(out, w.cpu()), k
    assert list(R.varispeed_batch_host([], 16)) == []
    with pytest.raises(ValueError):
        list(R.varispeed_batch_host([(np.arange(3.0), np.ones(2), np.zeros(10, np.float32))], 16))


def test_tapesync_project_on_reference_samples(par, tmp_path):
    """pytapesynch data flow on the reference's own demo data (fixtures under tests/golden/): rhythm+5percent.flac
    resampled with the lag curve of rhythm.tapesync must line up with rhythm.flac -- same length within a few
    hundred samples and a normalised correlation near 1 at (almost) zero lag, where the raw +5 % file has none."""
    import os
    import shutil
    from pyaudiorestoration_amd import io_ops
    from test_oracle_golden import GOLD
    src = str(tmp_path / "rhythm+5percent.flac")
    shutil.copy(os.path.join(GOLD, "rhythm+5percent.flac"), src)
    curve = par.pipeline.tapesync(os.path.join(GOLD, "rhythm.tapesync"), source=src)
    assert curve[-1, 1] > 1.4                                              # ~1.5 s of accumulated lag after 31.5 s
    y, sr, ch = io_ops.read_file(str(tmp_path / "rhythm+5percent_res.wav"))
    ref, sr_ref, _ = io_ops.read_file(os.path.join(GOLD, "rhythm.flac"))
    raw, _, _ = io_ops.read_file(src)
    assert sr == sr_ref == 44100 and ch == 1 and abs(len(y) - len(ref)) < 2000

    def best_corr(a, b, lo, hi, max_lag=400):
        """peak normalised correlation of a[lo+max_lag : hi-max_lag] against b shifted by -max_lag..+max_lag"""
        core = a[lo + max_lag:hi - max_lag, 0].astype(np.float64)
        cand = b[lo:hi, 0].astype(np.float64)
        dots = np.correlate(cand, core, mode="valid")                        # 2*max_lag + 1 lags
        energy = np.convolve(cand * cand, np.ones(len(core)), mode="valid")
        return float(np.max(np.abs(dots) / (np.linalg.norm(core) * np.sqrt(energy) + 1e-30)))
    for lo in (200000, 700000, 1200000):                                    # early, middle, late: the drift is gone
        synced = best_corr(ref, y, lo, lo + 40000)
        unsynced = best_corr(ref, raw, lo, lo + 40000)
        assert synced > 0.8 and synced > 3 * unsynced, (lo, synced, unsynced)


def test_config4_on_the_reference_dropout_sample(par, tmp_path):
    """BASELINE config 4 on the reference's own demo data (fixtures): dropouts_sample.flac + the 32 markers of
    dropouts_sample.drop -> device STFT 512/32, one-launch inpaint mask, fused ISTFT; full parity against the
    oracle's serial marker loop, dropouts lifted, everything outside the marker boxes untouched."""
    import json
    import os
    import shutil
    from oracle import oracle_np as O
    from pyaudiorestoration_amd import io_ops
    from test_oracle_golden import GOLD
    src = str(tmp_path / "dropouts_sample.flac")
    shutil.copy(os.path.join(GOLD, "dropouts_sample.flac"), src)
    proj = os.path.join(GOLD, "dropouts_sample.drop")
    healed = par.pipeline.heal_project(proj, source=src)
    x, sr, ch = io_ops.read_file(src)
    assert healed.shape == x.shape == (322531, 1) and sr == 44100
    y, _, _ = io_ops.read_file(str(tmp_path / "dropouts_sample_drops.wav"))
    assert np.array_equal(y, healed)
    cfg = json.load(open(proj))
    marks = [(m[0], m[1], m[2], m[3], cfg["surrounding"]) for m in cfg["dropouts"]]
    assert len(marks) == 32 and cfg["fft_size"] // cfg["fft_overlap"] == 32
    want = O.heal_dropouts(x, sr, marks, 512, 32)
    assert relerr(healed[:, 0], want[:, 0]) < TOL
    lifted = kept = 0
    touched = np.zeros(len(x), dtype=bool)
    for (t0, _, t1, _, _) in marks:
        a, b = int(min(t0, t1) * sr), int(max(t0, t1) * sr)
        touched[max(0, a - 600):b + 600] = True                       # box + the STFT frames that overlap it
        lifted += np.abs(healed[a:b, 0]).mean() > 1.05 * np.abs(x[a:b, 0]).mean()
        kept += np.abs(healed[a:b, 0]).mean() > 0.98 * np.abs(x[a:b, 0]).mean()
    assert lifted >= 16 and kept == 32                                 # the mask only ever boosts (gain >= 0 dB)
    assert relerr(healed[~touched, 0], x[~touched, 0]) < 1e-4         # the rest only sees the STFT round trip


def test_buffer_bound_ambiguity_is_settled_with_numpys_order(par):
    """int(mean(speeds) * span * 1.01): when the product sits within the device sum's uncertainty of an integer the
    bound is recomputed on the host in numpy's pairwise order (one copy of the speed samples; the device plan itself
    stands) -- positions equal the oracle's either way."""
    from oracle import oracle_c as C
    t = par.torch
    m, n = 400, 100000
    st = np.linspace(0, n, m)
    k = 100500                                                          # target integer for the bound
    sp = np.full(m, k / (n * 1.01))
    guess = np.mean(sp) * (st[-1] - st[0]) * 1.01
    assert abs(guess - round(guess)) < 1e-9
    info = {}
    pos = par.resampling.speed_to_pos_dev(t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), n, info=info).cpu().numpy()
    assert info["path"] == 0                                            # no serial re-plan for the bound alone
    ref, _ = C.speed_to_pos(st, sp, n)
    assert np.array_equal(pos, ref)
    sp2 = sp * 1.0001                                                   # a bound far from an integer: device scans
    info = {}
    pos2 = par.resampling.speed_to_pos_dev(t.from_numpy(st).cuda(), t.from_numpy(sp2).cuda(), n, info=info).cpu().numpy()
    assert info["path"] == 0
    ref2, _ = C.speed_to_pos(st, sp2, n)
    assert np.array_equal(pos2, ref2)


def test_sparse_speed_curves_take_the_chunked_exact_cumsum(par):
    """A curve with a handful of points makes segments of 10^5..10^7 samples.  Their sequential float64 cumsum is
    evaluated exactly in parallel (chunk maps in the parity-translation algebra, scan, verified final pass): the fused
    output must stay bit-identical to the position-array path, whose positions are bit-identical to the oracle."""
    from oracle import oracle_c as C
    t = par.torch
    R = par.resampling
    rng = np.random.default_rng(12)
    cases = [(3_000_000, [1.015, 1.015], 0.0), (3_000_000, [0.97, 1.04], 0.0), (2_500_000, [1.0, 0.5, 1.7], 0.0),
             (4_000_000, list(1.0 + 0.05 * np.sin(np.arange(11) * 0.9)), 123.25),
             (1_000_000, list(rng.uniform(0.8, 1.25, 40)), 0.0), (700_000, [1.0, 1.0], 0.0)]
    mixed = np.array([0, 100, 5000, 40000, 100000, 1_500_000, 1_500_010, 2_000_000], dtype=np.float64)
    cases.append((2_000_000, [1.1, 1.1, 1.1, 1.05, 1.0, 0.95, 1.0, 0.97], mixed))    # lane-per-segment and chunked segments mixed
    # (fast short segments: the reference sizes its output by the UNWEIGHTED mean speed and refuses curves that need more)
    for n, speeds, st0 in cases:
        sp = np.asarray(speeds, dtype=np.float64)
        st = st0 if isinstance(st0, np.ndarray) else np.linspace(0, n, len(sp)) + st0
        sig = rng.standard_normal(n).astype(np.float32)
        st_t, sp_t, sig_t = t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), t.from_numpy(sig).cuda()
        ref_pos, _ = C.speed_to_pos(st, sp, n)
        plan = R.speed_plan_dev(st_t, sp_t, n, fused=True)
        assert plan.fused_ok and plan.path == 0 and plan.len_out == len(ref_pos), (n, speeds[:3], plan.path, plan.len_out)
        pos_t = R.speed_to_pos_dev(st_t, sp_t, n)                        # block-parallel fill from the checkpoints
        assert t.equal(pos_t.cpu(), t.from_numpy(ref_pos))
        if n <= 1_000_000:                                               # and the lane-per-segment fill (slow here)
            from pyaudiorestoration_amd import _dev, _lib
            old = t.empty_like(pos_t)
            _lib.check(_lib.lib().par_speed_to_pos_fill(0, _dev.ptr(sp_t), plan.m, _dev.ptr(plan.work), _dev.ptr(old),
                                                         plan.len_out, _dev.stream_ptr(0)))
            assert t.equal(old, pos_t)
        for NT in (3, 32):
            out = R.varispeed_fused_dev(plan, sig_t, NT)
            assert same_resample(out, R.sinc_resample_dev(pos_t, sig_t, NT)), (n, speeds[:3], NT)
            oracle_spot(out, ref_pos, sig_t, NT, windows=4)


def test_sparse_curves_random_window_layouts(par):
    """Random curves of 2..7 points over 3..60 M samples: segments of 1..57 windows (4096 chunks of 256 steps each), power-
    of-two crossings of the running sum at arbitrary chunks, partial last windows and chunks -- positions bit for bit
    against the C oracle (whose cumsum is the plain sequential loop)."""
    from oracle import oracle_c as C
    t = par.torch
    rng = np.random.default_rng(2024)
    done = 0
    for case in range(40):
        n = int(rng.integers(3_000_000, 60_000_000))
        m = int(rng.integers(2, 8))
        cuts = np.sort(rng.uniform(0.05, 0.95, m - 2)) if m > 2 else np.zeros(0)
        st = np.concatenate(([0.0], cuts * n, [float(n)]))
        sp = rng.uniform(0.4, 2.5, m) if case % 3 else np.full(m, float(rng.uniform(0.5, 2.0)))
        try:
            ref, _ = C.speed_to_pos(st, sp, n)
        except ValueError:
            continue                                   # the reference's end_guess buffer is too small for this curve
        info = {}
        pos = par.resampling.speed_to_pos_dev(t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), n, info=info)
        assert info["path"] == 0 and t.equal(pos.cpu(), t.from_numpy(ref)), (case, n, m)
        del pos, ref
        done += 1
    assert done >= 15


def test_full_size_two_point_curve(par):
    """Config-2 length (345.6 M samples) with the sparsest possible curve -- a constant 1.5 % speed correction given as
    two points, i.e. ONE segment of 3.5e8 samples: the chunked exact cumsum and the block-parallel fill reproduce the C
    oracle's positions bit for bit, and the fused resampler reproduces the position-array one."""
    from oracle import oracle_c as C
    t = par.torch
    n = 345_600_000
    st, sp = np.array([0.0, float(n)]), np.array([1.015, 1.015])
    st_t, sp_t = t.from_numpy(st).cuda(), t.from_numpy(sp).cuda()
    info = {}
    pos_t = par.resampling.speed_to_pos_dev(st_t, sp_t, n, info=info)
    ref, _ = C.speed_to_pos(st, sp, n)
    assert info["path"] == 0 and t.equal(pos_t.cpu(), t.from_numpy(ref))
    del ref
    sig_t = t.empty(n, dtype=t.float32, device="cuda").normal_()
    plan = par.resampling.speed_plan_dev(st_t, sp_t, n, fused=True)
    assert plan.fused_ok and plan.len_out == pos_t.numel()
    out = par.resampling.varispeed_fused_dev(plan, sig_t, 32)
    assert same_resample(out, par.resampling.sinc_resample_dev(pos_t, sig_t, 32))
    oracle_spot(out, pos_t, sig_t, 32, windows=8)      # (pos_t equals the oracle's bit for bit: asserted above)
    del pos_t, plan, out
    # segments of 2.7e8 (258 windows of 4096 chunks: two steps of the window-level scan), 5e7, 2e7 and 5.6e6 samples in
    # one curve, positions against the C oracle
    st = np.array([0.0, 2.7e8, 3.2e8, 3.4e8, float(n)])
    sp = np.array([0.99, 0.99, 1.02, 1.03, 1.05])
    st_t, sp_t = t.from_numpy(st).cuda(), t.from_numpy(sp).cuda()
    pos_t = par.resampling.speed_to_pos_dev(st_t, sp_t, n, info=info)
    ref, _ = C.speed_to_pos(st, sp, n)
    assert info["path"] == 0 and t.equal(pos_t.cpu(), t.from_numpy(ref))


def test_zero_crossing_compaction_sizes(par):
    """K_track's sign-change compaction (count per tile, scan of tile counts in chunks of 1024, ordered write) against
    numpy for sizes around every boundary, incl. > 1024 tiles (multi-chunk scan), no crossings and all crossings."""
    W = par.wow
    t = par.torch
    rng = np.random.default_rng(21)
    for n in (2, 3, 255, 1023, 1024, 1025, 4096, 1024 * 1024 + 3, 3_000_001):
        x = rng.standard_normal(n)
        x[rng.integers(0, n, max(1, n // 50))] = 0.0                      # exact zeros count as "not positive"
        want = np.where(np.bitwise_xor(x[1:] > 0, x[:-1] > 0))[0]
        got = W.zero_crossings_dev(t.from_numpy(x).cuda()).cpu().numpy()
        assert got.dtype == np.int64 and np.array_equal(got, want), n
    assert W.zero_crossings_dev(t.ones(5000, dtype=t.float64, device="cuda")).numel() == 0
    alt = t.tensor([1.0, -1.0] * 3000, dtype=t.float64, device="cuda")
    assert np.array_equal(W.zero_crossings_dev(alt).cpu().numpy(), np.arange(5999))
    assert W.zero_crossings_dev(t.ones(1, dtype=t.float64, device="cuda")).numel() == 0
    assert np.array_equal(W.zero_crossings(np.array([0.5, -0.5, -0.1, 0.2])), [0, 2])


def test_library_is_reentrant_across_host_threads_and_streams(par):
    """SURVEY 8b threading contract: called from several host threads at once (QThreads in the GUI, one driver thread
    per GPU), each on its own stream, the library keeps no cross-call state -- every thread gets the serial result."""
    import threading
    t = par.torch
    R = par.resampling
    rng = np.random.default_rng(77)
    jobs = []
    for k in range(6):
        n = int(rng.choice([150000, 400000, 900000]))
        m = n // 256
        st = np.linspace(0, n, m)
        sp = 1.0 + 0.02 * np.sin(np.arange(m) * 0.05 + k)
        sig = rng.standard_normal(n).astype(np.float32)
        jobs.append((t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), t.from_numpy(sig).cuda(), n, 8 + 8 * (k % 4)))
    want = []
    for st_t, sp_t, sig_t, n, NT in jobs:
        plan = R.speed_plan_dev(st_t, sp_t, n, fused=True)
        want.append((R.varispeed_fused_dev(plan, sig_t, NT).clone(), par.fourier.get_mag(sig_t, 1024, 256).clone()))
    t.cuda.synchronize()
    got = [None] * len(jobs)
    errors = []

    def work(i):
        try:
            st_t, sp_t, sig_t, n, NT = jobs[i]
            with t.cuda.stream(t.cuda.Stream()):
                for _ in range(3):                                          # interleave plans, resamples and STFTs
                    plan = R.speed_plan_dev(st_t, sp_t, n, fused=True)
                    y = R.varispeed_fused_dev(plan, sig_t, NT)
                    mag = par.fourier.get_mag(sig_t, 1024, 256)
                t.cuda.current_stream().synchronize()
                got[i] = (y, mag)
        except Exception as e:                                              # surfaced below
            errors.append((i, repr(e)))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for (y, mag), (wy, wm) in zip(got, want):
        assert t.equal(y, wy) and t.equal(mag, wm)


def test_bench_contract_line():
    """bench.py prints ONE JSON line with the contract's keys, a roofline and a cpu_baseline object."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--seconds", "20", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_mean", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "archive_value", "archive_ms_per_file"):
        assert k in r, k
    assert r["archive_value"] == r["secondary_config5"]["batched_Msamples/s"] > 1000           # the base of the N-GPU curve
    assert 0 < r["cpu_baseline"]["speed_to_pos_share"] < 1 and r["cpu_baseline"]["sinc_only_value"] >= r["cpu_baseline"]["value"]
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["value"] > 1000 and "workload" in r["config"] and "model" not in r["config"]
    rl = r["roofline"]
    assert rl["bound"] == "hbm" and rl["peak"] == 8000.0 and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-4
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    # the core count is the one this process can use, the reported one rides along, and the per-core rate is on
    # the line (N threads ~ per-core x usable cores: the interpolator scales, speed_to_pos does not)
    assert 1 <= cb["cores"] == cb["cores_usable"] <= cb["cores_reported"] and cb["per_core_Msamples/s"] > 0.1
    assert 0.2 < cb["sinc_only_parallel_efficiency"] < 1.6, cb
    # the line names the kernel(s) of a timed launch; PMC side files are only quoted when they describe exactly those
    assert rl["kernel_symbols"] == ["k_sinc_pipe<1, 2>", "k_sinc_pipe<1, 1>", "k_sinc_fused_list"] and rl["warm_launches_before_timed"] >= 30
    assert "float16" in r["dtype"] and "mfma" in r["dtype"].lower()
    assert r["config"]["plan"].startswith("lazy")
    if rl["traffic"] is not None:
        assert rl["source"]["kernel_symbols"] == rl["kernel_symbols"]
    if "roofline_valu" in r:
        assert r["roofline_valu"]["kernel_symbols"] == rl["kernel_symbols"]


def test_headless_cli_respeed_and_resample(par, golden, tmp_path):
    """The headless pyrespeeder CLI on the reference's flutter_192.flac: same result as the golden config-3 flow."""
    import json
    import os
    import shutil
    from pyaudiorestoration_amd import cli, io_ops
    from test_oracle_golden import GOLD
    g = golden["samples"]
    f = str(tmp_path / "tape.flac")
    shutil.copy(os.path.join(GOLD, "flutter_192.flac"), f)
    assert cli.main(["respeed", "--trail", "0.2,4000,4.0,4000", "--quality", "32", f]) == 0
    y, sr, ch = io_ops.read_file(str(tmp_path / "tape_res.wav"))
    assert sr == 192000 and ch == 1 and len(y) == int(g["c3_len_pos"])
    chain_parity(y[5::8, 0], g["c3_y_dense"], g["c3_y_peak"], "cli respeed", max_over=3)
    curve = np.load(str(tmp_path / "tape_speed.npy"))
    assert relerr(curve[:, 1], g["c3_curve"][:, 1]) < 1e-7
    json.dump(g["c3_curve"].tolist(), open(str(tmp_path / "c.json"), "w"))
    assert cli.main(["resample", "--curve", str(tmp_path / "c.json"), "--quality", "32", "--suffix", "_b", f]) == 0
    y2, _, _ = io_ops.read_file(str(tmp_path / "tape_res_b.wav"))
    assert relerr(y2[g["c3_sel"], 0], g["c3_y_sel"]) < TOL
    # several files in one call: the next one is decoded on a helper thread while the current one is on the GPU
    more = []
    for k in range(3):
        more.append(str(tmp_path / f"copy{k}.flac"))
        shutil.copy(os.path.join(GOLD, "flutter_192.flac"), more[-1])
    bad = str(tmp_path / "broken.flac")
    open(bad, "wb").write(b"fLaC" + bytes(100))
    assert cli.main(["resample", "--curve", str(tmp_path / "c.json"), "--quality", "32", *more, bad]) == 1   # one failed
    for k in range(3):
        yk, _, _ = io_ops.read_file(str(tmp_path / f"copy{k}_res.wav"))
        assert np.array_equal(yk, y2)
    # constant correction = a two-point curve (one 8e5-sample segment: the chunked exact cumsum): against the C oracle
    from oracle import oracle_c as C
    assert cli.main(["resample", "--speed", "1.015", "--quality", "32", "--suffix", "_c", f]) == 0
    y3, _, _ = io_ops.read_file(str(tmp_path / "tape_res_c.wav"))
    x, _, _ = io_ops.read_file(f)
    pos, _ = C.speed_to_pos(np.array([0.0, len(x) / 192000]) * 192000, np.array([1.015, 1.015]), len(x))   # run(): t * sr
    ref3 = C.sinc(pos, np.ascontiguousarray(x[:, 0]), 32, threads=8)
    assert len(y3) == len(ref3) and relerr(y3[:, 0], ref3) < TOL


def test_headless_cli_respeed_project(par, golden, tmp_path):
    """8(f)-4 (r04): `cli respeed --project tape.spd` -- a saved pyrespeeder project with two traces, and one with a sine
    regression on top (then the regressed curve is the one that resamples, pyrespeeder_gui.py:133-140).  The curves against
    the ones the reference's own marker classes made of the same files (1e-7), the output against the reference's with its
    exact curve (1e-5), and the CLI end to end."""
    import json
    import os
    import shutil
    from pyaudiorestoration_amd import cli, io_ops, pipeline
    from test_oracle_golden import GOLD
    g = golden["spd"]
    f = str(tmp_path / "tape.flac")
    shutil.copy(os.path.join(GOLD, "flutter_192.flac"), f)
    x, sr, _ = io_ops.read_file(f)
    for tag in ("traces", "reg"):
        prj = os.path.join(GOLD, f"flutter_192_{tag}.spd")
        cfg = json.load(open(prj))
        curve = pipeline.project_speed_curve(cfg, len(x) / sr, sr)
        assert curve.shape == g[tag + "_curve"].shape and relerr(curve[:, 1], g[tag + "_curve"][:, 1]) < 1e-7, tag
        # the reference's exact curve through resampling.run
        par.resampling.run((f,), signal_data=((x, sr),), speed_curve=np.array(g[tag + "_curve"]), resampling_mode="Sinc",
                           sinc_quality=32, suffix="_exact_" + tag)
        y, _, _ = io_ops.read_file(str(tmp_path / f"tape_res_exact_{tag}.wav"))
        assert len(y) == int(g[tag + "_len_pos"]) and relerr(y[g[tag + "_sel"], 0], g[tag + "_y_sel"]) < TOL, tag
        # the CLI: project -> curve -> output
        assert cli.main(["respeed", "--project", prj, "--suffix", "_" + tag, f]) == 0
        y2, sr2, ch2 = io_ops.read_file(str(tmp_path / f"tape_res_{tag}.wav"))
        assert sr2 == sr and ch2 == 1 and len(y2) == int(g[tag + "_len_pos"])
        chain_parity(y2[g[tag + "_sel"], 0], g[tag + "_y_sel"], float(np.max(np.abs(g[tag + "_y_sel"]))), "project " + tag, max_over=12, q=0.999)
    # a project without markers is refused, a negative regression amplitude is a phase of pi (RegLine.__init__)
    with pytest.raises(ValueError):
        pipeline.project_speed_curve({"fft_size": 1024, "fft_overlap": 4, "lines": [], "regs": []}, 1.0, sr)
    r = list(map(float, g["reg"]))
    a = pipeline.master_reg_curve([r], len(x) / sr, sr, 256)
    b = pipeline.master_reg_curve([[r[0], r[1], -r[2], r[3], r[4] - np.pi, r[5]]], len(x) / sr, sr, 256)
    assert relerr(a[:, 1], b[:, 1]) < 1e-12


# --------------------------------------------------------------------------------- round-2 additions
def test_stereo_kernel_against_the_oracle(par, golden, tmp_path):
    """The stereo K_sinc form checked against the REFERENCE's arithmetic directly (not against the mono kernel):
    both channels of interleaved files against the C oracle on the oracle's own positions, for the specialised tap
    counts (32, 50) and a generic one, over fc == 1 and fc < 1 stretches; and resampling.run's channel-pairing branch
    (util/resampling.py:225-227) on a 2-channel Sinc-mode file against the reference-generated golden."""
    from oracle import oracle_c as C
    from pyaudiorestoration_amd import io_ops
    t = par.torch
    R = par.resampling
    rng = np.random.default_rng(77)
    for n, depth, NT in ((300000, 0.01, 32), (120000, 0.03, 50), (90000, 0.2, 12), (40000, 0.0, 32)):
        m = max(3, n // 256)
        st = np.linspace(0, n, m)
        sp = 1.0 + depth * np.sin(np.arange(m) * 0.05 + 0.3) + 0.0005 * rng.standard_normal(m)
        sig = rng.standard_normal((n, 2)).astype(np.float32)
        pos, _ = C.speed_to_pos(st, sp, n)
        plan = R.speed_plan_dev(t.from_numpy(st).cuda(), t.from_numpy(sp).cuda(), n, fused=True)
        assert plan.fused_ok and plan.len_out == len(pos)
        inter = t.from_numpy(sig).cuda()
        out = t.empty((plan.len_out, 2), dtype=t.float32, device="cuda")
        R.varispeed_fused_stereo_dev(plan, inter.reshape(-1)[0:], inter.reshape(-1)[1:], NT, out.reshape(-1)[0:],
                                     out.reshape(-1)[1:], sig_stride=2, len_in=n, out_stride=2)
        got = out.cpu().numpy()
        for c in (0, 1):
            want = C.sinc(pos, np.ascontiguousarray(sig[:, c]), NT)
            assert relerr(got[:, c], want) < TOL, (n, depth, NT, c)
    # run(): both channels selected -> one stereo launch; channel 0 is the golden's signal, channel 1 goes to the oracle
    g, gp = golden["sinc"], golden["speed_to_pos"]
    sr = 48000
    sig = np.stack((inputs.bench_signal(0, 96000, sr), inputs.noise(96000, 1)), axis=-1)
    curve = inputs.bench_speed_curve(2.0, sr)
    fn = str(tmp_path / "pair.flac")
    R.run((fn,), signal_data=((sig, sr),), speed_curve=curve, resampling_mode="Sinc", sinc_quality=32)
    y, sr2, ch = io_ops.read_file(str(tmp_path / "pair_res.wav"))
    assert sr2 == sr and ch == 2 and y.shape == (len(gp["bench_pos"]), 2)
    assert relerr(y[:, 0], g["bench_y"]) < TOL
    assert relerr(y[:, 1], C.sinc(gp["bench_pos"], np.ascontiguousarray(sig[:, 1]), 32)) < TOL


def test_full_size_config5_work_item(par):
    """One work item of BASELINE config 5 at FULL size -- a 600-s 192 kHz STEREO file (2 x 115.2 M samples,
    interleaved), per-file seed and curve phase as SURVEY 8d -- through size-independent properties: positions
    bit-equal to the C oracle's over the whole file, 24 oracle windows per channel spread over the ten minutes, the
    two channels independent of each other (a NaN in one poisons exactly its own +-NT neighbourhood), and the stereo
    launch equal to one mono launch per channel."""
    from oracle import oracle_c as C
    from pyaudiorestoration_amd import _lib, _dev
    t = par.torch
    R = par.resampling
    sr, dur, file_idx = 192000, 600.0, 3
    n = int(sr * dur)
    L = _lib.lib()
    s0 = _dev.stream_ptr(0)
    sig = t.empty((n, 2), dtype=t.float32, device="cuda")
    mono = t.empty(n, dtype=t.float32, device="cuda")
    for c in range(2):
        _lib.check(L.par_synth_signal_f32(0, _dev.ptr(mono), 0, n, float(sr), 0x5EED ^ (2 * file_idx + c), s0))
        sig[:, c] = mono
    del mono
    m = int(dur * sr / 256)
    st_t = t.empty(m, dtype=t.float64, device="cuda")
    sp_t = t.empty(m, dtype=t.float64, device="cuda")
    _lib.check(L.par_synth_speed_curve_f64(0, _dev.ptr(st_t), _dev.ptr(sp_t), m, dur, float(sr), 0.01, 0.55, 0.7 + file_idx, s0))
    plan = R.speed_plan_dev(st_t, sp_t, n, fused=True)
    assert plan.fused_ok and plan.path == 0
    ref_pos, _ = C.speed_to_pos(st_t.cpu().numpy(), sp_t.cpu().numpy(), n)
    pos_t = R.speed_to_pos_dev(st_t, sp_t, n)
    assert plan.len_out == len(ref_pos) and t.equal(pos_t.cpu(), t.from_numpy(ref_pos))
    flat = sig.reshape(-1)
    out = t.empty((plan.len_out, 2), dtype=t.float32, device="cuda")
    R.varispeed_fused_stereo_dev(plan, flat[0:], flat[1:], 32, out.reshape(-1)[0:], out.reshape(-1)[1:], sig_stride=2, len_in=n,
                                 out_stride=2)
    for c in range(2):
        one = R.varispeed_fused_dev(plan, flat[c:], 32, sig_stride=2, len_in=n)
        assert relerr(out[:, c].cpu().numpy(), one.cpu().numpy()) < 2e-6, c
        del one
        for i in np.linspace(0, len(ref_pos) - 3000, 24).astype(np.int64):
            lo = max(0, int(ref_pos[i]) - 200)
            hi = min(n, int(ref_pos[i + 2000]) + 200)
            win = sig[lo:hi, c].contiguous().cpu().numpy()
            ref = C.sinc(ref_pos[i:i + 2001] - lo, win, 32)[:2000]
            if lo == 0:
                ref = C.sinc(ref_pos[i:i + 2001], win, 32)[:2000]
            assert relerr(out[i:i + 2000, c].cpu().numpy(), ref) < TOL, (c, i)
    bad = 77_000_000
    sig[bad, 1] = float("nan")
    out_n = t.empty_like(out)
    R.varispeed_fused_stereo_dev(plan, flat[0:], flat[1:], 32, out_n.reshape(-1)[0:], out_n.reshape(-1)[1:], sig_stride=2,
                                 len_in=n, out_stride=2)
    ind = t.round(pos_t).to(t.int64)
    expect = (ind - 32 <= bad) & (bad < ind + 32)
    assert 60 <= int(expect.sum()) <= 68
    assert t.equal(t.isnan(out_n[:, 1]), expect) and not bool(t.isnan(out_n[:, 0]).any())
    # elsewhere the NaN changes nothing, bit for bit -- except in the tiles around it, which the streaming kernel hands to the
    # block kernel (BOTH channels of them) when it meets input float16 cannot carry: same numbers to float32 rounding
    jn = int(t.nonzero(expect)[0])
    far = t.ones_like(expect)
    far[max(0, jn - 16384):jn + 16384] = False
    assert t.equal(out_n[:, 0][far], out[:, 0][far]) and t.equal(out_n[:, 1][far], out[:, 1][far])
    near = ~far & ~expect
    for c in range(2):
        assert float((out_n[:, c][near] - out[:, c][near]).abs().max()) <= FUSED_TOL * float(out[:, c].abs().max()), c


def test_config4_x256_tiles_one_launch(par):
    """BASELINE config 4 at its benchmarked size: the reference's dropouts_sample signal (padded to a whole number of
    hops) tiled x256 = 82.6 M samples with the project's 32 markers repeated per tile -- 8192 gain boxes in ONE K_heal
    launch, 2.58 M frames through K_stft and the fused ISTFT.  Frames and boxes of every tile are the first tile's
    shifted by a whole number of frames, so every tile's interior must equal the oracle's output for ONE tile
    (dropout_healer_gui.py:111-166 restated in oracle_np)."""
    import json
    import os
    from oracle import oracle_np as O
    from pyaudiorestoration_amd import io_ops
    from test_oracle_golden import GOLD
    x, sr, _ = io_ops.read_file(os.path.join(GOLD, "dropouts_sample.flac"))
    cfg = json.load(open(os.path.join(GOLD, "dropouts_sample.drop")))
    n_fft, hop, tiles = 512, 32, 256
    n1 = -(-len(x) // hop) * hop                                     # 322 560 = 10 080 hops
    x1 = np.zeros(n1, dtype=np.float32)
    x1[:len(x)] = x[:, 0]
    marks = [(m[0], m[1], m[2], m[3], cfg["surrounding"]) for m in cfg["dropouts"]]
    geo1 = [par.pipeline.marker_geometry(mk, sr, hop, n_fft) for mk in marks]
    all_marks = []
    for k in range(tiles):
        off = k * n1 / sr
        for (a0, a1, b0, b1, s), g1 in zip(marks, geo1):
            mk = (a0 + off, a1, b0 + off, b1, s)
            gk = par.pipeline.marker_geometry(mk, sr, hop, n_fft)
            shift = k * (n1 // hop)                                       # same box, a whole number of frames on
            assert gk[0] - g1[0] == shift and gk[1] - g1[1] == shift and tuple(gk[2:]) == tuple(g1[2:]), (k, gk, g1)
            all_marks.append(mk)
    assert len(all_marks) == 8192
    want = O.heal_dropouts(x1, sr, marks, n_fft, hop)[:, 0]
    got = par.pipeline.heal_dropouts(np.tile(x1, tiles), sr, all_marks, n_fft, hop)[:, 0]
    assert got.shape == (n1 * tiles,)
    edge = 2 * n_fft                                                  # tile edges see the neighbour tile, not the reflection
    scale = np.max(np.abs(want))
    worst = 0.0
    for k in range(tiles):
        d = np.max(np.abs(got[k * n1 + edge:(k + 1) * n1 - edge] - want[edge:n1 - edge])) / scale
        worst = max(worst, d)
    assert worst < TOL, worst
    # r03: that call took the SPARSE path (only the frames a box can reach are transformed); the reference-shaped dense
    # path (transform and invert the whole 82.6 M-sample signal) gives the same signal
    plan = par.pipeline.heal_segments([par.pipeline.marker_geometry(mk, sr, hop, n_fft) for mk in all_marks],
                                      (n1 * tiles + n_fft // 2) // hop + 1, n1 * tiles + n_fft // 2, n1 * tiles, n_fft, hop)
    assert plan is not None and plan["total"] < 0.45 * n1 * tiles, plan and plan["total"]
    dense = par.pipeline.heal_dropouts(np.tile(x1, tiles), sr, all_marks, n_fft, hop, sparse=False)[:, 0]
    assert np.max(np.abs(dense - got)) / scale < 2e-6


def test_sparse_heal_equals_dense_and_oracle(par):
    """Sparse healing (r03) against the dense path and the oracle: boxes at the very start and end of the file (segments
    that keep the file's own reflect edge), overlapping and nested boxes (merged segments), far-apart boxes, stereo with a
    channel selection; and a marker set that covers most of the file, which must fall back to the dense path."""
    from oracle import oracle_np as O
    sr, n_fft, hop = 44100, 512, 32
    rng = np.random.default_rng(21)
    n = 400_000
    x = (0.4 * rng.standard_normal(n)).astype(np.float32)
    x[50_000:50_400] *= 0.05                                            # a few real dips
    x[200_000:200_900] *= 0.1
    dur = n / sr
    marks = [(0.012, 400.0, 0.030, 9000.0, 0.5),                       # surrounding frames reach frame 0..: file start
             (1.130, 500.0, 1.142, 8000.0, 0.5), (1.136, 300.0, 1.150, 6000.0, 1.0),        # overlapping
             (4.530, 800.0, 4.551, 12000.0, 0.5), (4.535, 2000.0, 4.540, 4000.0, 0.5),      # nested
             (7.000, 100.0, 7.004, 20000.0, 2.0),
             (dur - 0.040, 400.0, dur - 0.022, 9000.0, 0.5)]           # close to the end of the file
    plan = par.pipeline.heal_segments([par.pipeline.marker_geometry(m, sr, hop, n_fft) for m in marks],
                                      (n + n_fft // 2) // hop + 1, n + n_fft // 2, n, n_fft, hop)
    assert plan is not None and plan["segments"] == 5 and plan["total"] < 0.1 * n
    want = O.heal_dropouts(x, sr, marks, n_fft, hop)[:, 0]
    sparse = par.pipeline.heal_dropouts(x, sr, marks, n_fft, hop, sparse=True)[:, 0]
    dense = par.pipeline.heal_dropouts(x, sr, marks, n_fft, hop, sparse=False)[:, 0]
    scale = np.max(np.abs(want))
    assert np.max(np.abs(dense - want)) / scale < TOL and np.max(np.abs(sparse - want)) / scale < TOL
    assert np.max(np.abs(sparse - dense)) / scale < 2e-6
    assert np.max(np.abs(sparse - x)) / scale > 1e-2                    # the boxes did change something
    st = np.stack((x, x[::-1]), axis=1)
    got2 = par.pipeline.heal_dropouts(st, sr, marks, n_fft, hop, channels=(1,), sparse=True)
    want2 = O.heal_dropouts(np.ascontiguousarray(x[::-1]), sr, marks, n_fft, hop)[:, 0]
    assert np.max(np.abs(got2[:, 1] - want2)) / np.max(np.abs(want2)) < TOL
    # markers all over the file: the plan declines and the dense path runs
    many = [(t, 500.0, t + 0.01, 9000.0, 0.5) for t in np.arange(0.05, dur - 0.1, 0.03)]
    assert par.pipeline.heal_segments([par.pipeline.marker_geometry(m, sr, hop, n_fft) for m in many],
                                      (n + n_fft // 2) // hop + 1, n + n_fft // 2, n, n_fft, hop) is None
    got3 = par.pipeline.heal_dropouts(x, sr, many, n_fft, hop)[:, 0]
    assert np.max(np.abs(got3 - O.heal_dropouts(x, sr, many, n_fft, hop)[:, 0])) / scale < TOL


def test_heal_gain_kernel_band_shapes(par):
    """The gain kernels split a box into (bin, frame lane) threads: bands wider than one 256-bin chunk (2048-point frames,
    100 Hz .. 20 kHz = 925 bins: four chunks, the last one partial), a 3-bin band (85 frame lanes, most of them without a
    surrounding frame to sum), a one-frame box and overlapping boxes of different widths, dense and sparse, against the
    oracle's serial marker loop."""
    from oracle import oracle_np as O
    sr = 44100
    rng = np.random.default_rng(33)
    n = 260_000
    x = (0.3 * rng.standard_normal(n)).astype(np.float32)
    for a, b in ((40_000, 40_700), (120_000, 120_090), (200_000, 201_500)):
        x[a:b] *= 0.03
    for n_fft, hop in ((2048, 128), (512, 32)):
        marks = [(40_000 / sr, 100.0, 40_700 / sr, 20000.0, 0.5),           # wide band
                 (120_000 / sr, 1000.0, 120_090 / sr, 1000.0 + 2.6 * sr / n_fft, 1.0),      # ~3 bins, a frame or two wide
                 (200_000 / sr, 300.0, 201_500 / sr, 12000.0, 0.5), (200_400 / sr, 2000.0, 200_900 / sr, 2300.0, 2.0)]
        want = O.heal_dropouts(x, sr, marks, n_fft, hop)[:, 0]
        scale = np.max(np.abs(want))
        for sparse in (False, True):
            got = par.pipeline.heal_dropouts(x, sr, marks, n_fft, hop, sparse=sparse)[:, 0]
            assert np.max(np.abs(got - want)) / scale < TOL, (n_fft, sparse)
        assert np.max(np.abs(want - x)) / scale > 1e-2


@pytest.mark.parametrize("n_fft,hop,zp", [(16384, 4096, 1), (65536, 16384, 1), (4096, 1024, 4), (32768, 5000, 2), (1048576, 262144, 1),
                                           (32768, 8192, 1), (8192, 3000, 4), (1048576, 524288, 2), (131072, 40000, 4)])
def test_stft_above_8192_four_step(par, n_fft, hop, zp):
    """FFT sizes above 8192 (the GUI offers up to 2^20, util/widgets.py:333-349): the 16384-point single-workgroup kernel and
    the four-step transform (from 32768 points on) against the C
    oracle (float64 FFT of the same float32 frames), complex and magnitude, frame counts exact; strided input."""
    from oracle import oracle_c as C
    import scipy.signal
    n = max(3 * n_fft + 12345, 400000)
    x = inputs.noise(n, 21) + inputs.sine(n, 1234.5, 96000, 0.5)
    win = scipy.signal.get_window("blackmanharris", n_fft).astype(np.float32)
    want = C.stft(x, n_fft, hop, win, zp, mode=0, threads=8)
    got = par.fourier.stft(x, n_fft, hop, "blackmanharris", zp)
    assert got.shape == want.shape == (n_fft * zp // 2 + 1, n // hop + 1)
    assert relerr(got, want) < TOL
    mag = par.fourier.get_mag(x, n_fft, hop, "blackmanharris", zp)
    assert relerr(mag, np.abs(want) + 1e-7) < TOL
    st = np.stack((x, x[::-1]), axis=-1)
    xt = par.torch.from_numpy(st).cuda()
    got1 = par.fourier.stft(xt[:, 1], n_fft, hop, "blackmanharris", zp).cpu().numpy()
    assert relerr(got1, C.stft(np.ascontiguousarray(st[:, 1]), n_fft, hop, win, zp, mode=0, threads=8)) < TOL
    if n_fft <= 65536:                         # a signal that starts 4 bytes off the 8-byte grid: the unpaired loads of the column pass
        xc = par.torch.from_numpy(x).cuda()
        got2 = par.fourier.get_mag(xc[1:], n_fft, hop, "blackmanharris", zp).cpu().numpy()
        assert relerr(got2, np.abs(C.stft(np.ascontiguousarray(x[1:]), n_fft, hop, win, zp, mode=0, threads=8)) + 1e-7) < TOL


@pytest.mark.parametrize("n_fft,hop,zp", [(262144, 100000, 16), (1048576, 524288, 8), (1048576, 700001, 16)])
def test_stft_frames_of_2_22_to_2_24_points(par, n_fft, hop, zp):
    """The GUI's largest combinations (FFT size up to 2^20 x zero-padding up to 16, util/widgets.py:334-351): frames of
    2^22, 2^23 and 2^24 points go through one more radix step around the four-step transform (k_huge_gather / k_huge_out)
    instead of falling through to the reference's CPU chain (r02).  Complex and magnitude against the C oracle's float64
    FFT of the same float32 frames; the first and last frames hang over the ends of the signal (reflect boundary)."""
    from oracle import oracle_c as C
    import scipy.signal
    n = 2 * n_fft + 54321
    x = inputs.noise(n, 23) + inputs.sine(n, 4321.5, 192000, 0.5)
    win = scipy.signal.get_window("blackmanharris", n_fft).astype(np.float32)
    want = C.stft(x, n_fft, hop, win, zp, mode=0, threads=8)
    got = par.fourier.stft(x, n_fft, hop, "blackmanharris", zp)
    assert got.shape == want.shape == (n_fft * zp // 2 + 1, n // hop + 1)
    assert relerr(got, want) < TOL
    mag = par.fourier.get_mag(par.torch.from_numpy(x).cuda(), n_fft, hop, "blackmanharris", zp).cpu().numpy()
    assert relerr(mag, np.abs(want) + 1e-7) < TOL


@pytest.mark.parametrize("n_fft,hop,zp", [(16384, 4096, 1), (4096, 1000, 4), (16384, 16391, 1)])
def test_four_step_entry_point_at_its_smallest_size(par, n_fft, hop, zp):
    """par_stft_big_f32 accepts n_fft*zeropad = 16384 (fourier.stft sends that size to the single-workgroup kernel, so only a
    direct C-ABI caller reaches the 128 x 64 split: 64-point row pass, one wave per workgroup): complex and magnitude
    against the C oracle."""
    import scipy.signal
    from oracle import oracle_c as C
    from pyaudiorestoration_amd import _dev, _lib
    torch = par.torch
    L = _lib.lib()
    n = 150001
    x = inputs.noise(n, 5) + inputs.sine(n, 997.0, 96000, 0.7)
    win = scipy.signal.get_window("hann", n_fft).astype(np.float32)
    want = C.stft(x, n_fft, hop, win, zp, mode=0, threads=8)
    xt, wt = torch.from_numpy(x).cuda(), torch.from_numpy(win).cuda()
    frames, bins = int(L.par_stft_frames(n, n_fft, hop)), n_fft * zp // 2 + 1
    assert want.shape == (bins, frames)
    nbytes = int(L.par_stft_big_scratch_bytes(n, n_fft, hop, zp))
    assert nbytes == min(frames, (1 << 30) // (8192 * 8)) * 8192 * 8
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    for mode, dtype in ((0, torch.complex64), (1, torch.float32)):
        out = torch.empty((frames, bins), dtype=dtype, device="cuda")
        _lib.check(L.par_stft_big_f32(0, _dev.ptr(xt), n, 1, n_fft, hop, zp, _dev.ptr(wt), _dev.ptr(out), mode, _dev.ptr(scratch), nbytes,
                                      _dev.stream_ptr(0)))
        got = out.T.cpu().numpy()
        assert relerr(got, want if mode == 0 else np.abs(want) + 1e-7) < TOL, (mode, relerr(got, want if mode == 0 else np.abs(want) + 1e-7))
    # a scratch one byte short is refused, not overrun
    rc = L.par_stft_big_f32(0, _dev.ptr(xt), n, 1, n_fft, hop, zp, _dev.ptr(wt), _dev.ptr(out), 1, _dev.ptr(scratch), nbytes - 1, _dev.stream_ptr(0))
    assert rc != 0


def test_two_rank_config5_bench_flow(par):
    """`python bench.py --gpus 2` end to end on this box, started WITHOUT a launcher (bench.py re-executes itself under
    torch.distributed.run; both ranks share GPU 0, which needs the explicit PAR_OVERSUBSCRIBE=1): gloo rendezvous, the
    shared work queue, the stereo batch pipeline, the same-workload one-GPU base, the host-gather leg and the
    reductions -- the flow the driver launches on 2/4/8 GPUs -- with a small archive."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--files", "12",
           "--ring", "2", "--n1-files", "6", "--n1-e2e-files", "4"]
    refused = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    if par.torch.cuda.device_count() < 2:                   # fewer devices than ranks: a loud refusal, not a fold
        assert refused.returncode != 0 and "refusing to fold" in refused.stderr
        out = subprocess.run(cmd, env=dict(env, PAR_OVERSUBSCRIBE="1"), capture_output=True, text=True, timeout=900, cwd=root)
    else:
        out = refused
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["config"]["files"] == 12
    assert r["distinct_devices"] == min(2, par.torch.cuda.device_count())
    assert r["config"]["channel_samples_per_step"] > 12 * 2 * 115_000_000 and r["value"] > 1000.0
    lo, hi = r["config"]["files_per_rank_min_max"]
    assert 0 <= lo <= hi <= 12
    # the three keys VERDICT r02 asked for, and their arithmetic
    assert r["n1_same_workload_value"] > 1000.0
    assert abs(r["speedup_vs_n1"] - r["value"] / r["n1_same_workload_value"]) < 2e-3
    assert abs(r["efficiency"] - r["speedup_vs_n1"] / 2) < 1e-3
    assert r["value_e2e"] > 100.0 and r["e2e"]["n1_same_workload_value"] > 100.0 and r["value_e2e"] < r["value"]


def test_eight_rank_config5_flow_on_one_device(par):
    """Multi-GPU pre-flight (no 8-GPU node has run this code yet): the REAL 8-rank flow of `bench.py --gpus 8`
    -- gloo rendezvous, the shared TCP-store work queue, eight batch drivers with their planner threads, the reductions -- on
    this box's device(s), with short files so that eight processes fit.  Every file of the archive is processed exactly once
    and no rank is starved."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["PAR_OVERSUBSCRIBE"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--files", "64",
           "--c5-seconds", "20", "--ring", "2", "--n1-files", "8", "--no-e2e"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["n_gpus"] == 8 and r["scaling"] == "strong" and r["config"]["files"] == 64
    assert r["config"]["files_processed"] == 64                      # every file once: nothing dropped, nothing done twice
    lo, hi = r["config"]["files_per_rank_min_max"]
    assert 0 <= lo <= hi <= 64 and hi >= 8                           # (the queue hands out four files per request)
    assert r["config"]["channel_samples_per_step"] > 64 * 2 * 3_800_000 and r["value"] > 100.0
    assert r["distinct_devices"] == min(8, par.torch.cuda.device_count())


def test_batch_gather_equals_batch_dev(par):
    """resampling.varispeed_batch_gather (the host-gather leg): pinned host results equal the device-resident batch's,
    in order, across ring wrap-arounds, for mono and interleaved stereo items of different lengths."""
    t = par.torch
    sr = 48000
    items = []
    for k, seconds in enumerate((1.0, 0.7, 1.3, 0.9, 1.1, 0.5, 0.8)):
        n = int(sr * seconds)
        curve = inputs.bench_speed_curve(seconds, sr, phase=0.7 + k)
        st = t.from_numpy(curve[:, 0] * sr).cuda()
        sp = t.from_numpy(np.ascontiguousarray(curve[:, 1])).cuda()
        sig = inputs.bench_signal(k, n, sr)
        if k % 2:
            sig = np.stack([sig, inputs.bench_signal(100 + k, n, sr)], axis=1)
        items.append((st, sp, t.from_numpy(np.ascontiguousarray(sig)).cuda()))
    want = [(o.cpu().numpy().copy(), p.len_out) for _, o, p in par.resampling.varispeed_batch_dev(items, 32)]
    got = []
    for k, host, plan in par.resampling.varispeed_batch_gather(items, 32):
        assert host.is_pinned() and k == len(got)
        got.append((host.numpy().copy(), plan.len_out))
    assert len(got) == len(want)
    for (a, la), (b, lb) in zip(got, want):
        assert la == lb and a.shape == b.shape and np.array_equal(a, b)


def test_partials_tracker_piptrack(par):
    """PartialsTracker (util/wow_detection.py:361-387): the device piptrack against the oracle's restatement of librosa's
    algorithm on the SAME magnitudes (parity of the restatement itself is unpinned: librosa is not in the reference
    checkout), and the tracker's contract -- freqs stay the drawn trail, the maps land in .pitches / .magnitudes."""
    from oracle import oracle_np as O
    sr, n, n_fft, hop = 48000, 72000, 1024, 256
    x = inputs.pilot(n, sr)
    t = par.torch
    mag = par.fourier.get_mag(t.from_numpy(x).cuda(), n_fft, hop, "hann", 1)
    S = (mag.cpu().numpy().astype(np.float64) - 1e-7) * np.sqrt(n_fft)
    for fmin, fmax, thr in ((3000.0, 5000.0, 0.15), (0.0, sr, 0.01), (150.0, 4000.0, 0.5)):
        p_t, m_t = par.wow.piptrack_dev(mag, n_fft, sr, fmin, fmax, thr)
        want_p, want_m = O.piptrack(S.astype(np.float32), sr, n_fft, fmin, fmax, thr)
        got_p, got_m = p_t.cpu().numpy(), m_t.cpu().numpy()
        # the device rebuilds S from mag in float32: a bin within one rounding of the threshold or of a tie may flip
        same = (got_p != 0) == (want_p != 0)
        assert same.mean() > 0.9999, (fmin, fmax, thr, same.mean())
        assert np.max(np.abs(got_p - want_p)[same]) < 0.05 and np.max(np.abs(got_m - want_m)[same]) < 1e-4 * want_m.max()
    trail = [(0.2, 3950.0), (1.3, 4080.0)]
    tr = par.wow.wow_detectors["Partials"](mag, x[:, None], list(trail), n_fft, hop, sr, 0.5, "Linear")
    base = par.wow.wow_detectors["Freehand Draw"](mag, x[:, None], list(trail), n_fft, hop, sr, 0.5, "Linear")
    assert np.array_equal(tr.freqs, base.freqs) and np.array_equal(tr.times, base.times)
    assert tuple(tr.pitches.shape) == tuple(mag.shape)
    pk = tr.pitches.cpu().numpy()
    hits = pk[:, 40:-40]                                      # the pilot's 4 kHz line, flutter included
    assert ((hits > 3900) & (hits < 4100)).sum(axis=0).min() == 1
    # the edge frames are those of librosa >= 0.10's zero-padded centred STFT, not K_stft's reflected ones
    xz = np.concatenate((np.zeros(n_fft // 2, np.float32), x, np.zeros(n_fft // 2, np.float32)))
    Sz = (O.get_mag(xz, n_fft, hop, "hann")[:, (n_fft // 2) // hop:][:, :mag.shape[1]] - 1e-7) * np.sqrt(n_fft)
    wp, _ = O.piptrack(Sz.astype(np.float32), sr, n_fft, 3950.0, 4080.0, 0.15)
    for fr in (0, 1, mag.shape[1] - 1):
        assert np.array_equal(pk[:, fr] != 0, wp[:, fr] != 0) and np.max(np.abs(pk[:, fr] - wp[:, fr])) < 0.05, fr


def test_find_delay_rival_peaks(par):
    """the float32 transform locates the correlation peak to ~1e-6 of its height; when another peak sits within
    that distance of the top more than two lags away (a repeated passage; the +/- lobes under ignore_phase) the reference's
    float64 argmax may pick the other one, a whole passage away.  The rivals are re-evaluated exactly: a second copy of the
    template that is 3e-7 LOUDER than the first must win, an equal one must lose to the lower lag, and under ignore_phase
    an inverted, 3e-7 louder copy must win."""
    from oracle import oracle_np as O
    rng = np.random.default_rng(33)
    T = rng.standard_normal(4000)
    for scale2, ign in ((1.0 + 3e-7, False), (1.0 - 3e-7, False), (-(1.0 + 3e-7), True)):
        a = 1e-9 * rng.standard_normal(60000)                  # (a louder floor would decide between the peaks itself)
        a[10000:14000] += T
        a[30517:34517] += scale2 * T
        b = np.zeros(60000)
        b[28000:32000] = T
        want_d, want_c = O.find_delay(a.copy(), b.copy(), ignore_phase=ign)
        got_d, got_c = par.correlation.find_delay(a.copy(), b.copy(), ignore_phase=ign)
        assert abs(got_d - want_d) < 1e-6 and abs(got_c - want_c) < 1e-9, (scale2, ign, got_d, want_d)
        assert abs(want_d - (-18000 if abs(scale2) < 1 else 2517)) < 1.0             # the test does distinguish the two passages


def test_correlation_of_long_windows(par):
    """Windows longer than one 2^20-point transform can hold (2.7 s at 192 kHz) are correlated section pair by section
    pair: full correlation against scipy's FFT method, delay and height against the oracle's find_delay."""
    import scipy.signal
    from oracle import oracle_np as O
    rng = np.random.default_rng(21)
    na, nb = 1_300_000, 1_100_000
    base = np.convolve(rng.standard_normal(na + 5000), np.hanning(15) / 7, mode="same")
    a = base[2500:2500 + na].copy()
    b = 0.7 * np.interp(np.arange(nb) + 2500 - 811.4, np.arange(len(base)), base) + 0.02 * rng.standard_normal(nb)
    full = par.correlation.xcorr(a, b, mode="full")
    an, bn = a / np.linalg.norm(a), b / np.linalg.norm(b)
    want = scipy.signal.correlate(an, bn, mode="full", method="fft")
    assert full.shape == want.shape and np.max(np.abs(full - want)) < 5e-6
    m = 1_050_000                                           # equal lengths: the delay is then the plain lag
    got = par.correlation.find_delay(a[:m].copy(), b[:m].copy())
    ref = O.find_delay(a[:m].copy(), b[:m].copy())
    assert abs(got[0] - ref[0]) < 1e-5 and abs(got[1] - ref[1]) < 1e-8 and abs(abs(got[0]) - 811.4) < 0.2, (got, ref)


def test_correlate_sources_flow(par):
    """pytapesynch_gui.py:108-133: band-pass both windows, then find_delay -- against the oracle's filter + find_delay
    (each pinned to the reference's goldens on its own), with and without a window / phase."""
    from oracle import oracle_np as O
    from pyaudiorestoration_amd import pipeline
    sr, n = 48000, 24000
    rng = np.random.default_rng(11)
    base = np.convolve(rng.standard_normal(n + 400), np.hanning(9) / 4, mode="same")
    ref = base[200:200 + n].copy()
    src = 0.8 * np.interp(np.arange(n) + 200 - 37.3, np.arange(len(base)), base) + 0.01 * rng.standard_normal(n)
    for window_name, ignore_phase in ((None, False), ("hann", False), ("hann", True)):
        want_d, want_c = O.find_delay(O.butter_bandpass_filter(ref.copy(), 300.0, 6000.0, sr, order=3),
                                      O.butter_bandpass_filter(src.copy(), 300.0, 6000.0, sr, order=3),
                                      ignore_phase=ignore_phase, window_name=window_name)
        keep = ref.copy()
        got_d, got_c = pipeline.correlate_sources(ref, src, sr, 300.0, 6000.0, ignore_phase, window_name)
        assert abs(got_d * sr - want_d) < 1e-5 and abs(got_c - want_c) < 1e-8, (window_name, got_d * sr, want_d, got_c, want_c)
        assert np.array_equal(ref, keep)
    assert abs(abs(got_d * sr) - 37.3) < 0.1


def test_correlation_on_the_device(par, golden):
    """util/correlation.py on the device: xcorr through one complex four-step FFT (float32: 1e-6 of the peak),
    find_delay with its peak neighbourhood re-evaluated in float64 (the reference-generated golden to 1e-9), window
    applied in place like the reference, every scipy mode, tape-sync sized windows against scipy, ignore_phase, and
    the reference's IndexError for a peak on the last lag."""
    import scipy.signal
    from pyaudiorestoration_amd import correlation as C
    g = golden["correlation"]
    aa = np.sin(np.arange(521) * 1.0)
    bb = np.sin(np.arange(521) * 1.0 + 3)
    keep = aa.copy()
    assert np.allclose(C.find_delay(aa, bb, window_name="hann"), g["find_delay"], rtol=1e-9, atol=1e-9)
    assert not np.array_equal(aa, keep)                                  # windowed in place, like the reference
    a, b = inputs.noise(200, 40).astype(np.float64), inputs.noise(200, 41).astype(np.float64)
    assert np.max(np.abs(C.xcorr(a, b, mode="same") - g["xcorr_same"])) < 2e-6
    for n, m in ((7, 7), (64, 9), (9, 64), (33, 1), (5000, 3000)):
        u, v = inputs.noise(n, n).astype(np.float64), inputs.noise(m, m + 1).astype(np.float64)
        un, vn = u / np.linalg.norm(u), v / np.linalg.norm(v)
        for mode in ("full", "same") + (("valid",) if n >= m else ()):
            assert np.max(np.abs(C.xcorr(u, v, mode) - scipy.signal.correlate(un, vn, mode=mode))) < 2e-6, (n, m, mode)
    # tape-sync sized: two band-limited takes, one delayed by a fractional number of samples, one inverted
    sr, n = 44100, 300000
    rng = np.random.default_rng(5)
    src = scipy.signal.sosfiltfilt(scipy.signal.butter(3, [200 / (sr / 2), 3000 / (sr / 2)], btype="band", output="sos"),
                                   rng.standard_normal(n + 4000))
    t = np.arange(n)
    ref = src[2000:2000 + n].copy()
    other = np.interp(t + 2000 - 37.3, np.arange(len(src)), src)

    def ref_find_delay(x, y, ignore_phase):
        res = scipy.signal.correlate(x / np.linalg.norm(x), y / np.linalg.norm(y), mode="same", method="fft")
        k = int(np.argmax(np.abs(res) if ignore_phase else res))
        xv = 0.5 * (res[k - 1] - res[k + 1]) / (res[k - 1] - 2 * res[k] + res[k + 1]) + k
        return xv - len(res) // 2, res[k] - 0.25 * (res[k - 1] - res[k + 1]) * (xv - k)
    for y, ign in ((other, False), (-other, True)):
        d, c = C.find_delay(ref.copy(), y.copy(), ignore_phase=ign)
        wd, wc = ref_find_delay(ref, y, ign)
        assert abs(d - wd) < 1e-6 and abs(c - wc) < 1e-9, (d, wd, c, wc)
        assert abs(abs(d) - 37.3) < 0.05
    with pytest.raises(IndexError):
        C.find_delay(np.array([0.0, 0.0, 0.0, 1.0]), np.array([0.0, 1.0]))   # peak on the last lag: parabolic() reads f[x+1]


# ---- lazy plans (r05; csrc/pos_plan.h) -----------------------------------------------------------------------------
def _plan_arrays(plan):
    """(header words, seg_start, seg_off, S) of a device plan (layout: csrc/pos_plan.h plan_view)."""
    import torch
    raw = plan.work.cpu().numpy()
    m = plan.m
    hdr = raw[:256].view(np.int32)
    off = 256
    seg_start = raw[off:off + 8 * m].view(np.int64)
    seg_off = raw[off + 8 * m:off + 16 * m].view(np.float64)
    S = raw[off + 16 * m:off + 24 * m].view(np.float64)
    return hdr, seg_start.copy(), seg_off.copy(), S.copy()


def _lazy_bound(n, smin):
    return (0.55 * n * n + 16.0 * n + 64.0) * 2.0 ** -53 / smin


def test_lazy_plan_equals_eager_plan(par):
    """A lazy plan (closed-form segment sums + exact sums for the candidates of the offset chain) must reproduce the eager
    plan's offsets, lengths and trim bit for bit; its closed-form sums must sit inside the bound the candidate test relies on;
    curves it does not vouch for must come back eager; and the fused resampler must produce the same output from either."""
    import torch
    from oracle import oracle_c as C
    R = par.resampling
    rng = np.random.default_rng(77)
    cases = []
    sc = inputs.bench_speed_curve(30.0, 192000)
    cases.append(("bench30", sc[:, 0] * 192000, sc[:, 1], int(192000 * 30.0), True))
    sc = inputs.bench_speed_curve(4.0, 48000, hop=64, depth=0.02, rate_hz=3.0)
    cases.append(("flutter", sc[:, 0] * 48000, sc[:, 1], int(48000 * 4.0), True))
    for scale, m, hop in ((1.0, 6000, 256), (0.07, 3000, 256), (50.0, 800, 256), (0.9, 5000, 64), (3.3, 2000, 1000), (1.0, 300, 7)):
        st = np.arange(m) * float(hop)
        sp = scale * (1.0 + 0.01 * np.sin(np.arange(m) * 0.013 + 0.4) + 1e-7 * rng.standard_normal(m))
        ok = 0.0625 <= sp.min() and sp.max() <= 64.0 and hop * sp.max() * 1.01 <= 1024
        cases.append((f"gentle{scale}/{hop}", st, sp, int(st[-1] * float(np.mean(sp)) * 0.97), ok))
    st = np.arange(4000) * 256.0
    sp = 1.0 + 0.01 * np.sin(np.arange(4000) * 0.013)
    sp2 = sp.copy()
    sp2[1234] *= 1.03                                                      # one steep segment: the whole plan goes eager
    cases.append(("one-steep", st, sp2, int(st[-1] * 0.97), False))
    cases.append(("no-trim", st, sp, int(st[-1] * 1.2), True))            # the trim never fires
    st5 = np.arange(40) * 5000.0                                           # segments of ~5000 outputs: too long for the closed form
    cases.append(("long-segs", st5, 1.0 + 0.002 * np.sin(np.arange(40) * 0.3), int(st5[-1] * 0.97), False))
    worst = 0.0
    n_cand_seen = 0
    for name, st, sp, n_in, want_lazy in cases:
        st_t, sp_t = torch.from_numpy(np.ascontiguousarray(st, dtype=np.float64)).cuda(), torch.from_numpy(np.ascontiguousarray(sp, dtype=np.float64)).cuda()
        eager = R.speed_plan_dev(st_t, sp_t, n_in, fused=True, eager=True)
        he, ss_e, so_e, S_e = _plan_arrays(eager)
        lazy = R.speed_plan_dev(st_t, sp_t, n_in, fused=True)
        hl, ss_l, so_l, S_l = _plan_arrays(lazy)
        assert eager.fused_ok and not eager.lazy and lazy.fused_ok, name
        assert lazy.lazy == want_lazy, (name, lazy.lazy, hl[26:29])
        assert (lazy.len_out, lazy.trimmed, lazy.path) == (eager.len_out, eager.trimmed, eager.path), name
        assert np.array_equal(ss_e, ss_l), name
        nseg = lazy.m - 1
        assert np.array_equal(so_e.view(np.int64), so_l.view(np.int64)), (name, "offset chain differs from the eager plan's")
        ref, _ = C.speed_to_pos(np.asarray(st, dtype=np.float64), np.asarray(sp, dtype=np.float64), n_in)
        assert lazy.len_out == len(ref), name
        if lazy.lazy:
            n = np.diff(ss_l)[:nseg].astype(np.float64)
            smin = np.minimum(sp[:-1], sp[1:])
            ratio = np.abs(S_l[:nseg] - S_e[:nseg]) / _lazy_bound(n, smin)
            worst = max(worst, float(ratio.max()))
            assert ratio.max() < 0.5, (name, "closed-form sum outside half its bound", float(ratio.max()))
            n_cand_seen += int(hl[28])
            assert 0 <= hl[28] <= 65536 and hl[27] == 0, name
        sig = inputs.bench_signal(0, int(n_in), 48000)
        sig_t = torch.from_numpy(sig).cuda()
        for NT in (32, 50):
            a = R.varispeed_fused_dev(eager, sig_t, NT).cpu().numpy()
            b = R.varispeed_fused_dev(lazy, sig_t, NT).cpu().numpy()
            want = C.sinc(ref, sig, NT, threads=8)
            pk = float(np.max(np.abs(want)))
            assert np.max(np.abs(a - b)) <= FUSED_TOL * pk, (name, NT, "lazy vs eager output")
            assert np.max(np.abs(b - want)) <= TOL * pk, (name, NT)
    assert n_cand_seen > 0          # the exact-sum list was exercised
    print(f"lazy plans: worst |closed form - exact sum| / bound = {worst:.3f}, candidates seen {n_cand_seen}")


def test_lazy_plan_near_ties_take_the_exact_walk(par):
    """Window centres of a lazy plan's fused output against rint() of the oracle's exact positions, on curves built so that many
    positions land within 1e-9 of a half-integer (speed exactly 1 after a half-sample start offset is not possible through a
    curve, so: speed 2/3 and 2 give positions on thirds and halves), plus the bench curve, through the debug slot of the fused
    kernel's placement (the output of a unit-ramp signal x[i] = i IS the position to ~1e-3, so rint errors show as jumps)."""
    import torch
    from oracle import oracle_c as C
    R = par.resampling
    for speed, hop, m in ((2.0, 64, 3000), (0.5, 256, 2000), (2.0 / 3.0, 300, 1500), (1.0, 256, 3000)):
        st = np.arange(m) * float(hop)
        sp = np.full(m, speed)
        sp[1::2] *= 1.0 + 2.0 ** -30                                       # not a constant curve: segments differ by an ulp-ish ramp
        n_in = int(st[-1] * speed * 0.98)
        st_t, sp_t = torch.from_numpy(st).cuda(), torch.from_numpy(sp).cuda()
        lazy = R.speed_plan_dev(st_t, sp_t, n_in, fused=True)
        eager = R.speed_plan_dev(st_t, sp_t, n_in, fused=True, eager=True)
        assert lazy.fused_ok and lazy.lazy, speed
        ref, _ = C.speed_to_pos(st, sp, n_in)
        assert lazy.len_out == len(ref) == eager.len_out
        sig = inputs.noise(n_in, seed=5)
        sig_t = torch.from_numpy(sig).cuda()
        want = C.sinc(ref, sig, 32, threads=8)
        got = R.varispeed_fused_dev(lazy, sig_t, 32).cpu().numpy()
        pk = float(np.max(np.abs(want)))
        # a wrong window centre on white noise moves the output by ~1e-1 of the peak at half-integer positions
        assert np.max(np.abs(got - want)) <= TOL * pk, (speed, float(np.max(np.abs(got - want)) / pk))


# ---- 8(f)3, second half (r05): batched K_sosfiltfilt and the heuristic dropout repair ----------------------------------------
def test_sosfiltfilt_batch_equals_single_calls_and_scipy(par):
    """par_sosfiltfilt_batch_f64: every row of a batch equals the single-signal call bit for bit and scipy to 1e-9; one cascade
    for all rows or one per row (the bands of a multi-band analysis); scipy's too-short-input error; and the point of it -- 64
    bands x 10^6 samples in one call against the loop of single calls (>= 20x)."""
    import time
    import scipy.signal
    t = par.torch
    F = par.filters
    rng = np.random.default_rng(31)
    n, n_sig, fs = 50_001, 7, 48000.0
    x = rng.standard_normal((n_sig, n))
    x_t = t.from_numpy(x).cuda()
    sos = scipy.signal.butter(3, [300 / (fs / 2), 3000 / (fs / 2)], btype="band", output="sos")
    y = F.sosfiltfilt_batch_dev(sos, x_t)
    for i in range(n_sig):
        assert t.equal(y[i], F.sosfiltfilt_dev(sos, x_t[i].contiguous())), i
        assert relerr(y[i].cpu().numpy(), scipy.signal.sosfiltfilt(sos, x[i])) < 1e-9
    lows = np.geomspace(100, 8000, n_sig)
    many = np.stack([scipy.signal.butter(3, [lo / (fs / 2), 1.5 * lo / (fs / 2)], btype="band", output="sos") for lo in lows])
    y = F.sosfiltfilt_batch_dev(many, x_t)
    for i in range(n_sig):
        assert t.equal(y[i], F.sosfiltfilt_dev(many[i], x_t[i].contiguous())), i
    yb = F.bandpass_batch_dev(x_t, lows, 1.5 * lows, fs, order=3)
    assert t.equal(yb, y)
    with pytest.raises(ValueError, match="greater than padlen"):
        F.sosfiltfilt_batch_dev(sos, x_t[:, :10].contiguous())
    with pytest.raises(ValueError):
        F.sosfiltfilt_batch_dev(many[:3], x_t)
    # 64 bands x 10^6 samples
    n, n_sig = 1_000_000, 64
    x_t = t.from_numpy(rng.standard_normal((n_sig, n))).cuda()
    lows = np.geomspace(50, 15000, n_sig)
    many = np.stack([scipy.signal.butter(3, [lo / (fs / 2), 1.3 * lo / (fs / 2)], btype="band", output="sos") for lo in lows])

    def timed(fn, reps=3):
        fn()
        t.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        t.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    rows = [x_t[i].contiguous() for i in range(n_sig)]
    t_loop = timed(lambda: [F.sosfiltfilt_dev(many[i], rows[i]) for i in range(n_sig)])
    t_batch = timed(lambda: F.sosfiltfilt_batch_dev(many, x_t))
    print(f"sosfiltfilt 64 bands x 1e6 samples: loop {t_loop * 1e3:.2f} ms, batched {t_batch * 1e3:.2f} ms ({t_loop / t_batch:.1f}x)")
    # (measured 11.6x: 58.5 -> 5.0 ms, 1.8 TB/s over the six section passes; the loop is launch-bound.  A wall-clock ratio is not a
    # unit test's business on a shared GPU: reported above and in NOTES, asserted only loosely)
    assert t_loop / t_batch >= 2.0
    # the VALUES of this size: n_sig x L >= 20 M takes the LDS-tiled batched kernels (k_sos_block_zero_bt / _run_bt), which the
    # 7 x 50 001 case above never reaches
    yb = F.sosfiltfilt_batch_dev(many, x_t)
    for i in (0, 1, 17, 40, 63):
        assert t.equal(yb[i], F.sosfiltfilt_dev(many[i], rows[i])), i
    for i in (0, 31, 63):
        assert relerr(yb[i].cpu().numpy(), scipy.signal.sosfiltfilt(many[i], x_t[i].cpu().numpy())) < 1e-9, i


def test_heuristic_repair_against_the_reference(par):
    """pipeline.heal_heuristic against dropouts_gui.MainWindow.process_heuristic itself (tests/golden/heuristic.npz): the low-rate
    two-channel tape (all three bands alive) and the GUI's defaults on dropouts_sample.flac; numpy in -> numpy out, device
    tensor in -> device tensor out, the input untouched."""
    import os
    from test_oracle_golden import GOLD
    from pyaudiorestoration_amd import io_ops
    g = np.load(os.path.join(GOLD, "heuristic.npz"))
    sig = inputs.heuristic_input()
    keep = sig.copy()
    sr, fft, hop, mw, ms, nb, bf, fu, fl = g["low_params"]
    y = par.pipeline.heal_heuristic(sig, int(sr), int(fft), int(hop), float(mw), float(ms), int(nb), float(bf), float(fu), float(fl))
    assert isinstance(y, np.ndarray) and y.dtype == np.float32 and y.shape == sig.shape and np.array_equal(sig, keep)
    want = g["low"]
    assert float(np.max(np.abs(want - sig))) > 0.1
    assert relerr(y, want) < TOL, relerr(y, want)
    # relative to the CHANGE the repair makes, not only to the signal's peak
    assert float(np.max(np.abs(y - want))) < 1e-4 * float(np.max(np.abs(want - sig)))
    yt = par.pipeline.heal_heuristic(par.torch.from_numpy(sig).cuda(), int(sr), int(fft), int(hop), float(mw), float(ms), int(nb),
                                     float(bf), float(fu), float(fl))
    assert par.torch.is_tensor(yt) and np.array_equal(yt.cpu().numpy(), y)
    x, sr_d, _ = io_ops.read_file(os.path.join(GOLD, "dropouts_sample.flac"))
    y = par.pipeline.heal_heuristic(x, sr_d, 512, 64)
    assert float(np.max(np.abs(y[::3] - g["gui_every3"]))) < TOL * float(g["gui_peak"])


def test_staged_host_transfers_are_exact(par):
    """_dev.to_dev / _dev.to_host move large pageable numpy arrays through a ring of pinned chunks filled / drained by threads (r06):
    byte-exact for sizes around the chunk and ring boundaries, for the widening upload, into a caller's array, and when two
    threads ask at once (the second one takes the plain copy)."""
    import threading
    t = par.torch
    from pyaudiorestoration_amd import _dev
    rng = np.random.default_rng(12)
    C, R = _dev._STAGE_CHUNK, _dev._STAGE_RING
    for nbytes in (_dev._STAGE_MIN, _dev._STAGE_MIN + 4, C * R + 8, C * (R + 1) - 4, C * (2 * R + 3) + 12):
        a = rng.integers(0, 2 ** 31, nbytes // 4, dtype=np.int32).view(np.float32)
        d = _dev.to_dev(a, t.float32, 0)
        assert d.dtype == t.float32 and np.array_equal(d.view(t.int32).cpu().numpy(), a.view(np.int32)), nbytes
        back = _dev.to_host(d)
        assert back.dtype == np.float32 and np.array_equal(back.view(np.int32), a.view(np.int32)), nbytes
        into = np.empty_like(a)
        assert _dev.to_host(d, into) is into and np.array_equal(into.view(np.int32), a.view(np.int32))
    a16 = rng.integers(-30000, 30000, 40_000_001, dtype=np.int16)            # widening: the narrow form travels
    assert np.array_equal(_dev.to_dev(a16, t.float32, 0).cpu().numpy(), a16.astype(np.float32))
    a64 = rng.standard_normal(9_000_001)
    assert np.array_equal(_dev.to_host(_dev.to_dev(a64, t.float64, 0)), a64)
    a2 = rng.standard_normal((5_000_001, 4)).astype(np.float32)
    assert np.array_equal(_dev.to_host(_dev.to_dev(a2[:, ::2], t.float32, 0)), a2[:, ::2])     # strided source, 2-D
    tv = t.arange(30_000_000, dtype=t.float32, device="cuda").reshape(6_000_000, 5).T      # a transposed view comes back as one
    hv = _dev.to_host(tv)
    assert hv.shape == (5, 6_000_000) and hv.T.flags.c_contiguous and np.array_equal(hv, tv.cpu().numpy())
    # column views of an interleaved file (the reference's signal[:, ch] / output[:, k]): gathered and scattered on the staging threads
    inter = rng.standard_normal((12_000_001, 2)).astype(np.float32)
    assert np.array_equal(_dev.to_dev(inter[:, 1], t.float32, 0).cpu().numpy(), inter[:, 1])
    assert np.array_equal(_dev.contiguous(inter[:, 0], np.float64), inter[:, 0].astype(np.float64))
    back2 = np.zeros_like(inter)
    _dev.host_assign(back2[:, 1], inter[:, 0])
    assert np.array_equal(back2[:, 1], inter[:, 0]) and not back2[:, 0].any()
    res = [None, None]

    def up(k):
        res[k] = _dev.to_dev(a64 + k, t.float64, 0).cpu().numpy()
    th = [threading.Thread(target=up, args=(k,)) for k in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert np.array_equal(res[0], a64) and np.array_equal(res[1], a64 + 1)
