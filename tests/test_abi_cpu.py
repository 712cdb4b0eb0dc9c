"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/par_hip.h declares, argument validation works without a GPU, and the product package never
reaches into oracle/ nor silently falls back to a CPU path."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "par_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(par_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pyaudiorestoration_amd import _lib
    L = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/par_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes signature table and header disagree"
    assert L.par_version() >= 106
    # ... and nothing else: the product build exports no entry point the header does not declare (an experiment
    # hook was exported undeclared)
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if re.search(r" [TW] par_[a-z0-9_]+$", ln)})
    assert exported == syms, sorted(set(exported) ^ set(syms))


def test_documents_name_the_abi_version_the_library_reports():
    """DESIGN / INTEGRATION / the header speak of the ABI version par_version() returns (a bump must reach the documents)"""
    from pyaudiorestoration_amd import _lib
    v = str(_lib.lib().par_version())
    for name in ("DESIGN.md", "INTEGRATION.md", os.path.join("include", "par_hip.h")):
        assert ("ABI " + v in open(os.path.join(ROOT, name)).read() or "version " + v in open(os.path.join(ROOT, name)).read()), name


def test_product_sources_read_no_environment_knobs():
    """experiment knobs (getenv, timing #ifs) are compiled only into -DPAR_EXPERIMENT builds"""
    csrc = os.path.join(ROOT, "pyaudiorestoration_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".h")):
            continue
        depth_exp = 0
        for ln in open(os.path.join(csrc, name)):
            t = ln.strip()
            if t.startswith("#ifdef PAR_EXPERIMENT"):
                depth_exp += 1
            elif depth_exp and t.startswith(("#else", "#endif")):
                depth_exp -= 1
            elif "getenv(" in t and not t.startswith("//"):
                assert depth_exp, f"{name}: getenv outside PAR_EXPERIMENT: {t}"


def test_pure_host_entry_points_and_argument_errors():
    from pyaudiorestoration_amd import _lib
    L = _lib.lib()
    # integer frame arithmetic is exact: C1, C3, C4 sizes
    assert L.par_stft_frames(186291, 1024, 256) == 728
    assert L.par_stft_frames(811063, 1024, 256) == 3169
    assert L.par_stft_frames(322531 + 256, 512, 32) == 10088
    assert L.par_speed_plan_bytes(1000) >= 1000 * 24
    # null pointers are rejected before any HIP call
    rc = L.par_sinc_resample_f32(0, None, 10, None, 1, 10, 32, None, 1, None)
    assert rc == 1 and "null" in _lib.last_error()
    rc = L.par_stft_f32(0, ctypes.c_void_p(8), 100, 1, 1000, 10, 1, ctypes.c_void_p(8), ctypes.c_void_p(8), 0, 0, None)
    assert rc == 3 and "power of two" in _lib.last_error()      # PAR_ERR_UNSUPPORTED -> caller falls through
    with pytest.raises(_lib.ParUnsupported):
        _lib.check(rc)
    # the multi-file launch (ABI 106): an empty batch is a no-op, a bad item is named before any HIP call
    assert L.par_varispeed_fused_batch_f32(0, 0, None, 32, None) == 0
    assert L.par_varispeed_fused_batch_f32(0, 2, None, 32, None) == 1 and "null items" in _lib.last_error()
    arr = (_lib.FusedItem * 2)()
    for f in arr:
        f.speeds, f.work, f.aux, f.sig0, f.out0 = 8, 8, 8, 8, 8
        f.m, f.max_out, f.len_out, f.sig_stride, f.len_in, f.out_stride = 4, 100, 50, 1, 60, 1
    arr[1].len_out = 101
    assert L.par_varispeed_fused_batch_f32(0, 2, arr, 32, None) == 1 and "item 1" in _lib.last_error() and "len_out" in _lib.last_error()
    arr[1].len_out, arr[1].sig1 = 50, 8                            # a second input channel without a second output
    assert L.par_varispeed_fused_batch_f32(0, 2, arr, 32, None) == 1 and "sig1 / out1" in _lib.last_error()
    # scratch sizing (host arithmetic): the fused ISTFT needs no frame array up to 2048 points when its overlap-add span fits
    # LDS; 4096 and 8192 points and over-long hops go through [n_frames][n_fft]
    assert L.par_istft_scratch_floats(100, 512, 32) == 0 and L.par_istft_scratch_floats(100, 2048, 512) == 0
    assert L.par_istft_scratch_floats(100, 4096, 1024) == 100 * 4096 and L.par_istft_scratch_floats(100, 8192, 2048) == 100 * 8192
    assert L.par_istft_scratch_floats(100, 128, 5000) == 100 * 128
    # four-step STFT: one H-point complex array per frame of a batch, at most 1 GiB (and at least 16 frames)
    assert L.par_stft_big_scratch_bytes(1000, 1024, 256, 1) == 0                      # single-workgroup sizes need none
    n = 57_600_000
    frames = L.par_stft_frames(n, 65536, 16384)
    assert L.par_stft_big_scratch_bytes(n, 65536, 16384, 1) == frames * 32768 * 8      # 3516 frames fit the cap
    assert L.par_stft_big_scratch_bytes(n, 32768, 64, 1) == (1 << 30)                  # capped: 8192 frames of 128 KiB
    assert L.par_stft_big_scratch_bytes(n * 40, 1 << 20, 4096, 2) == 128 * (1 << 20) * 8


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pyaudiorestoration_amd import fourier, resampling, filters
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fourier.stft(np.zeros(4096, dtype=np.float32), 1024, 256)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        resampling.sinc_wrapper(np.arange(10.0), np.zeros(10, dtype=np.float32), 0, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        filters.butter_bandpass_filter(np.zeros(100), 0, 20, 172.0, order=3)


def test_cli_refuses_to_run_without_a_gpu(tmp_path):
    """The headless tool has no CPU path either: without a ROCm GPU it stops before touching a file; its argument
    contract (--curve xor --speed) is enforced by the parser."""
    import torch
    from pyaudiorestoration_amd import cli
    with pytest.raises(SystemExit):
        cli.main(["resample", "--curve", "c.json", "--speed", "1.01", "x.wav"])      # mutually exclusive
    with pytest.raises(SystemExit):
        cli.main(["resample", "x.wav"])                                              # one of them is required
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(SystemExit, match="no CPU fallback"):
        cli.main(["resample", "--speed", "1.015", str(tmp_path / "missing.wav")])


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "pyaudiorestoration_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower() or f == "__init__.py" and False, f"{f} mentions the oracle"


def test_registry_and_signatures_match_reference_names():
    import inspect
    from pyaudiorestoration_amd import fourier, resampling, wow_detection, filters, correlation
    assert list(inspect.signature(fourier.stft).parameters)[:5] == ["x", "n_fft", "step", "window_name", "zeropad"]
    assert list(inspect.signature(fourier.hip_rfft2).parameters)[:5] == ["n_fft", "step", "window", "x", "zeropad"]
    assert list(inspect.signature(resampling.run).parameters) == [
        "filenames", "signal_data", "speed_curve", "resampling_mode", "sinc_quality", "use_channels", "prog_sig",
        "lag_curve", "suffix"]
    assert list(inspect.signature(resampling.sinc_wrapper_mt).parameters) == ["output", "sample_at", "signal", "lowpass", "NT"]
    assert list(inspect.signature(resampling.sinc_core).parameters) == ["sample_at", "signal", "lowpass", "output", "win_func", "N"]
    assert list(inspect.signature(resampling.speed_to_pos).parameters) == ["sampletimes", "speeds", "num_imput_samples"]
    assert list(inspect.signature(filters.butter_bandpass_filter).parameters) == ["data", "lowcut", "highcut", "fs", "order"]
    assert list(inspect.signature(wow_detection.Track.__init__).parameters)[1:9] == [
        "spectrum", "signal", "trail", "fft_size", "hop", "sr", "tolerance_st", "adaptation_mode"]
    assert set(wow_detection.wow_detectors) == {"Center of Gravity", "Peak", "Peak Track", "Zero-Crossing", "Partials",
                                                "Freehand Draw", "Correlation", "Sine Regression"}
    assert correlation.parabolic([1, 3, 2], 1) == (1.1666666666666667, 3.0416666666666665)     # KAT7


def test_wav_io_roundtrip(tmp_path):
    from pyaudiorestoration_amd import io_ops
    x = np.random.default_rng(0).standard_normal((1000, 2)).astype(np.float32)
    p = str(tmp_path / "a.wav")
    io_ops.write_wav_float(p, x, 192000)
    y, sr, ch = io_ops.read_file(p)
    assert sr == 192000 and ch == 2 and np.array_equal(x, y)
    io_ops.write_file(p, x[:, 0], 44100, 1, suffix="_res")
    y, sr, ch = io_ops.read_file(str(tmp_path / "a_res.wav"))
    assert ch == 1 and np.array_equal(y[:, 0], x[:, 0])
    # every sample format the reader takes (libsndfile's dtype="float32" scalings), an unknown chunk with an odd size in front of
    # the data, WAVE_FORMAT_EXTENSIBLE, and a data chunk that claims more than the file holds
    import struct
    rng = np.random.default_rng(1)

    def wav(tag, bits, ch, payload, extensible=False, junk=b"", claim=None):
        fmt = struct.pack("<HHIIHH", 0xFFFE if extensible else tag, ch, 48000, 48000 * ch * bits // 8, ch * bits // 8, bits)
        if extensible:
            fmt += struct.pack("<HHI", 22, bits, 3) + struct.pack("<H", tag) + bytes(14)
        body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt
        if junk:
            body += b"LIST" + struct.pack("<I", len(junk)) + junk + (b"\0" if len(junk) & 1 else b"")
        body += b"data" + struct.pack("<I", len(payload) if claim is None else claim) + payload
        q = str(tmp_path / "f.wav")
        with open(q, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
        return io_ops.read_file(q)
    i16 = rng.integers(-32768, 32767, (500, 2), dtype=np.int16)
    y, sr, ch = wav(1, 16, 2, i16.tobytes(), junk=b"odd")
    assert sr == 48000 and ch == 2 and y.dtype == np.float32 and np.array_equal(y, i16.astype(np.float32) / 32768.0)
    i32 = rng.integers(-2 ** 31, 2 ** 31 - 1, (300, 1), dtype=np.int32)
    assert np.array_equal(wav(1, 32, 1, i32.tobytes())[0], (i32.astype(np.float64) / 2147483648.0).astype(np.float32))
    i24 = rng.integers(-2 ** 23, 2 ** 23 - 1, (400, 3), dtype=np.int32)
    b24 = np.stack([(i24 >> s) & 0xFF for s in (0, 8, 16)], axis=-1).astype(np.uint8).tobytes()
    assert np.array_equal(wav(1, 24, 3, b24, extensible=True)[0], (i24 / 8388608.0).astype(np.float32))
    f64 = rng.standard_normal((200, 2))
    assert np.array_equal(wav(3, 64, 2, f64.tobytes())[0], f64.astype(np.float32))
    y, _, _ = wav(3, 32, 2, x.tobytes(), claim=10 ** 9)
    assert np.array_equal(y, x)
    with pytest.raises(ValueError, match="unsupported"):
        wav(1, 8, 1, bytes(100))


def test_host_filter_design_matches_reference_branches():
    from pyaudiorestoration_amd import filters
    x = np.arange(10.0)
    assert filters._design(0, 0, 100, 3) is None          # pass-through branch returns data itself
    assert filters._design(10, 20, 100, 3).shape == (3, 6)  # band
    assert filters._design(10, 0, 100, 3).shape == (2, 6)   # high
    assert filters._design(0, 20, 100, 5).shape == (3, 6)   # low
    assert filters.make_odd(4) == 5 and filters.make_odd(5) == 5
    assert np.allclose(filters.moving_average(x, 3), [1, 2, 3, 4, 5, 6, 7, 8])


# ----------------------------------------------------------------- host logic of the tracker / correlation mirrors
def test_parabolic_kat(golden):
    """correlation.parabolic is host arithmetic (KAT7); xcorr / find_delay run on the device: tests/test_hip_parity.py."""
    from pyaudiorestoration_amd import correlation as C
    assert C.parabolic([1, 3, 2], 1) == (1.1666666666666667, 3.0416666666666665) == tuple(golden["correlation"]["parabolic"])


def test_tracker_host_geometry_and_correlation_core(golden):
    """sample_trail / band_to_bins reproduce the reference's trail sampling and Freehand trace; the spline matrix the
    device CorrelationTracker is fed equals scipy's interp1d(kind='quadratic') on the reference's grid; fit_sin
    recovers a planted sine."""
    import inputs
    from oracle import oracle_np as O
    from pyaudiorestoration_amd import wow_detection as W
    g = golden["trackers"]
    sr, n, n_fft, hop = (int(v) for v in g["cfg"])
    spec = O.get_mag(inputs.pilot(n, sr), n_fft, hop, "blackmanharris", 1)
    trail = [(1.3, 4000.0), (0.2, 4000.0)]                               # unsorted on purpose
    f0, f1, times, freqs = W.sample_trail(trail, spec.shape[1], hop, sr)
    assert trail == [(0.2, 4000.0), (1.3, 4000.0)]                       # sorted in place, like the reference
    assert np.array_equal(times, g["freehand_draw_times"]) and np.array_equal(freqs, g["freehand_draw_freqs"])
    lo, hi = W.band_to_bins(freqs.min(), freqs.max(), n_fft, sr, spec.shape[0])
    assert hi - lo >= 4
    import scipy.interpolate
    log_f = np.log2(O.fft_freqs(n_fft, sr)[lo:hi])
    M = W.correlation_spline_matrix(log_f)
    grid = np.linspace(log_f[0], log_f[-1], 4 * (hi - lo))
    for col in (3, 57):
        y = np.asarray(spec[lo:hi, col], dtype=np.float64)
        want = scipy.interpolate.interp1d(log_f, y, kind="quadratic")(grid)      # util/wow_detection.py:413-414
        assert M.shape == (4 * (hi - lo), hi - lo) and np.max(np.abs(M @ y - want)) < 1e-12 * np.max(np.abs(want))
    assert W.band_to_bins(10.0, 12.0, 1024, 44100, 513) == (-1, 3)       # widened symmetrically from (1, 1)
    assert set(W.wow_detectors) == {"Center of Gravity", "Peak", "Peak Track", "Zero-Crossing", "Partials",
                                    "Freehand Draw", "Correlation", "Sine Regression"}
    t = np.arange(0, 20, 0.05)
    fit = W.fit_sin(t, 0.3 * np.sin(2 * np.pi * 0.55 * t + 0.4) + 1.0, assumed_freq=0.55)
    assert abs(abs(fit["amp"]) - 0.3) < 1e-6 and abs(fit["freq"] - 0.55) < 1e-6 and abs(fit["offset"] - 1.0) < 1e-6
    curve = np.stack((t, 0.01 * np.sin(2 * np.pi * 0.55 * t) + 1.0), axis=-1)
    amp, omega, phase, zero = W.trace_sine_reg(curve, 2.0, 18.0, rpm="33.333")
    assert abs(abs(amp) - 0.01) < 1e-6 and abs(omega / (2 * np.pi) - 0.55) < 1e-5 and zero == 0


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


# ----------------------------------------------------------------------------------- native FLAC decoder (host code)
FLAC_CASES = [
    # name, frames, ch, bps, kwargs for tests/flac_writer.encode_flac
    ("mono16_fixed2", 20000, 1, 16, dict(blocksize=4096, kind="fixed", order=2)),
    ("mono16_fixed0_4", 9000, 1, 16, dict(blocksize=1152, kind="fixed", order=4, method=1, porder=3)),
    ("mono8_verbatim", 700, 1, 8, dict(blocksize=256, kind="verbatim")),
    ("mono16_constant", 3000, 1, 16, dict(blocksize=1024, kind="constant")),
    ("stereo16_midside_lpc", 30000, 2, 16, dict(blocksize=4096, stereo="mid_side", kind="lpc", order=3, porder=2)),
    ("stereo16_leftside", 12000, 2, 16, dict(blocksize=2048, stereo="left_side", kind="fixed", order=1)),
    ("stereo24_rightside_var", 15000, 2, 24, dict(blocksize=4096, stereo="right_side", kind="fixed", order=3, variable=True,
                                                  vary_blocks=True)),
    ("stereo24_fixedflag_varying_blocks", 15000, 2, 24, dict(blocksize=4096, kind="fixed", order=2, vary_blocks=True)),
    ("mono16_escape_wasted", 8000, 1, 16, dict(blocksize=1024, kind="fixed", order=2, escape_first=True, wasted=3)),
    ("quad20_lpc8", 6000, 4, 20, dict(blocksize=576, kind="lpc", order=8, method=1)),
    ("stereo32_midside", 5000, 2, 32, dict(blocksize=1024, stereo="mid_side", kind="fixed", order=2, method=1)),
]


def _flac_pcm(frames, ch, bps, seed, wasted=0, constant=False):
    rng = np.random.default_rng(seed)
    t = np.arange(frames)
    amp = (1 << (bps - 1)) * 0.4
    cols = []
    for c in range(ch):
        x = amp * np.sin(2 * np.pi * (0.01 + 0.003 * c) * t + c) + rng.normal(0, amp * 1e-3, frames)
        x = np.clip(np.rint(x), -(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int64)
        if constant:
            x[:] = x[0]
        cols.append((x >> wasted) << wasted)
    return np.stack(cols, axis=-1)


@pytest.mark.parametrize("name,frames,ch,bps,kw", FLAC_CASES, ids=[c[0] for c in FLAC_CASES])
def test_native_flac_decoder_every_subframe_and_stereo_mode(tmp_path, name, frames, ch, bps, kw):
    """libpar_hip.so's host FLAC decoder (frame-parallel, CRC-8/CRC-16/MD5 checked) against streams written by the
    test-side encoder, and against the independent pure-Python decoder."""
    import flac_writer
    from pyaudiorestoration_amd import io_ops
    pcm = _flac_pcm(frames, ch, bps, len(name), kw.get("wasted", 0), kw.get("kind") == "constant")
    path = tmp_path / (name + ".flac")
    path.write_bytes(flac_writer.encode_flac(pcm, 48000, bps, **kw))
    want = (pcm / float(1 << (bps - 1))).astype(np.float32)
    for threads in (1, 0):
        got, sr, n_ch = io_ops.read_flac(str(path), n_threads=threads)
        assert sr == 48000 and n_ch == ch and got.shape == want.shape and np.array_equal(got, want), (name, threads)
    ref, _, _ = io_ops.read_flac_py(str(path))
    assert np.array_equal(ref, want)


def test_native_flac_decoder_parallel_split_and_corruption(tmp_path):
    """A stream long enough to be cut into many worker ranges decodes identically with 1 and N threads; a flipped
    payload bit is caught by the frame CRC / MD5 (ParError), a truncated file too."""
    import flac_writer
    from pyaudiorestoration_amd import _lib, io_ops
    pcm = _flac_pcm(1_500_000, 2, 16, 99)
    blob = flac_writer.encode_flac(pcm, 192000, 16, blocksize=4096, stereo="mid_side", kind="fixed", order=2)
    p = tmp_path / "long.flac"
    p.write_bytes(blob)
    want = (pcm / 32768.0).astype(np.float32)
    a, sr, ch = io_ops.read_flac(str(p), n_threads=1)
    b, _, _ = io_ops.read_flac(str(p), n_threads=16)
    assert sr == 192000 and ch == 2 and np.array_equal(a, want) and np.array_equal(b, want)
    assert len(blob) > 16 * 65536                                   # really split into 16 ranges
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 0x10
    (tmp_path / "bad.flac").write_bytes(bytes(bad))
    with pytest.raises(_lib.ParError):
        io_ops.read_flac(str(tmp_path / "bad.flac"))
    (tmp_path / "short.flac").write_bytes(blob[:len(blob) // 3])
    with pytest.raises(_lib.ParError):
        io_ops.read_flac(str(tmp_path / "short.flac"))
    x, sr, ch = io_ops.read_file(str(p))                            # the reference-named entry point uses it
    assert x.shape == want.shape and sr == 192000 and ch == 2
    # a STREAMINFO that claims 2^35 samples for a 3 MB file is refused BEFORE anything is allocated for it
    huge = bytearray(blob)
    v = int.from_bytes(huge[18:26], "big")
    huge[18:26] = ((v & ~((1 << 36) - 1)) | (1 << 35)).to_bytes(8, "big")
    (tmp_path / "huge.flac").write_bytes(bytes(huge))
    with pytest.raises(ValueError, match="more than a"):
        io_ops.read_flac(str(tmp_path / "huge.flac"))


def test_lag_curve_from_tapesync_markers():
    """Headless LagLine (util/markers.py:730-790): two markers -> order-1 spline = the straight line through them,
    extrapolated; four markers -> a cubic through all of them; sampling grid and end point follow the marker rate."""
    import json
    from pyaudiorestoration_amd import pipeline
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "rhythm.tapesync")))
    sr, hop = 44100, cfg["fft_size"] // cfg["fft_overlap"]
    dur = 1344000 / sr
    curve = pipeline.lag_curve_from_markers(cfg["markers"], dur, sr, hop, cfg["smoothing"])
    (a0, _, b0, _, d0, _), (a1, _, b1, _, d1, _) = cfg["markers"]
    t0, t1 = (a0 + b0) / 2, (a1 + b1) / 2
    line = lambda t: d0 + (d1 - d0) * (t - t0) / (t1 - t0)
    end = abs(dur + line(dur))
    assert curve.shape == (int(end * sr / hop), 2) and curve[0, 0] == 0 and abs(curve[-1, 0] - end) < 1e-9
    assert np.allclose(curve[:, 1], line(curve[:, 0]), rtol=0, atol=1e-9)
    assert abs((d1 - d0) / (t1 - t0) - (1 - 1 / 1.05)) < 2e-4             # the sample really is 5 % fast
    marks = [(t, 0, t, 0, 0.01 * t * t, 0.5) for t in (1.0, 2.0, 4.0, 7.0)]
    cubic = pipeline.lag_curve_from_markers(marks, 10.0, 48000, 256, smoothing=3)
    for t in (1.0, 2.0, 4.0, 7.0):
        i = np.argmin(np.abs(cubic[:, 0] - t))
        assert abs(cubic[i, 1] - 0.01 * cubic[i, 0] ** 2) < 1e-9       # a cubic reproduces the parabola exactly
    one = pipeline.lag_curve_from_markers([(3.0, 0, 3.0, 0, 0.25, 1.0)], 10.0, 48000, 256)
    assert np.all(one[:, 1] == 0.25)


def test_pmc_summary_picks_the_timed_kernel():
    """tools/summarise_profiles.pick_counters: the counters quoted on the bench line are those of the kernel(s) the line says a
    timed launch consists of -- not of whichever K_sinc kernel sorts last (the opt-in moment kernel's rows had
    overwritten the timed block kernel's)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("summarise_profiles", os.path.join(root, "tools", "summarise_profiles.py"))
    S = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(S)
    rows = {("void par::k_sinc_fused<1, 32, 4>(long, float const*, float const*)", "FETCH_SIZE"): 1993364.0,
            ("void par::k_sinc_fused<1, 32, 4>(long, float const*, float const*)", "SQ_INSTS_VALU"): 3.281e9,
            ("void par::k_sinc_fused<2, 32, 4>(long, float const*, float const*)", "FETCH_SIZE"): 5.0e5,
            ("void par::k_sinc_pipe<1>(par::S2Args)", "FETCH_SIZE"): 1855254.0,
            ("void par::k_sinc_pipe<1>(par::S2Args)", "SQ_INSTS_VALU"): 1.944e9,
            ("par::k_sinc_fused_list(long, float const*, long)", "FETCH_SIZE"): 1000.0,
            ("par::k_sinc_fused_list(long, float const*, long)", "SQ_INSTS_VALU"): 1.0e6}
    v, missing = S.pick_counters(rows, ["k_sinc_fused<1, 32, 4>"])
    assert not missing and v == {"FETCH_SIZE": 1993364.0, "SQ_INSTS_VALU": 3.281e9}
    v, missing = S.pick_counters(rows, ["k_sinc_pipe<1>", "k_sinc_fused_list"])
    assert not missing and v["FETCH_SIZE"] == 1856254.0 and abs(v["SQ_INSTS_VALU"] - 1.945e9) < 1.0
    v, missing = S.pick_counters(rows, ["k_sinc_stream"])
    assert missing == ["k_sinc_stream"] and v == {}


def test_usable_cores_never_exceed_the_affinity_mask():
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    usable, reported = bench.usable_cores()
    assert 1 <= usable <= reported
    if hasattr(os, "sched_getaffinity"):
        assert usable <= len(os.sched_getaffinity(0))
