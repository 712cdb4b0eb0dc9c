#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (imported read-only).

TEST INFRASTRUCTURE ONLY.  Runs in the build container, where the reference
checkout exists at /root/reference; it never runs on the GPU box and nothing is
copied from the reference -- only arrays it computed are stored.

The reference needs numba and soundfile at import time; neither is installed, so
two throw-away stub modules are created in a temp dir: ``numba.jit`` becomes an
identity decorator (the jitted functions then run as the plain numpy/Python they
are written in -- same arithmetic, no fastmath re-association) and ``soundfile`` is
empty (no file I/O is exercised).  ``stft`` then falls through torch (no GPU) and
pyfftw (absent) to ``np_rfft_pick`` -- the "numpy CPU path" BASELINE.json names.

Usage:  python oracle/gen_golden.py [--ref /root/reference]
"""
import argparse
import logging
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs  # noqa: E402


def import_reference(ref):
    stubs = tempfile.mkdtemp(prefix="par_stubs_")
    os.makedirs(os.path.join(stubs, "numba"))
    with open(os.path.join(stubs, "numba", "__init__.py"), "w") as f:
        f.write("def jit(*a, **k):\n"
                "    if len(a) == 1 and callable(a[0]) and not k:\n"
                "        return a[0]\n"
                "    return lambda fn: fn\n")
    os.makedirs(os.path.join(stubs, "soundfile"))
    open(os.path.join(stubs, "soundfile", "__init__.py"), "w").close()
    sys.path.insert(0, stubs)
    sys.path.insert(0, ref)
    logging.disable(logging.CRITICAL)
    from util import fourier, resampling, filters, wow_detection, correlation
    return fourier, resampling, filters, wow_detection, correlation


def written_len(st, sp, buflen):
    """Length of the prefix speed_to_pos actually wrote (quirk 2: without the trim
    branch the reference returns its whole np.empty buffer, tail uninitialised).
    Uses the reference's own n_i recurrence (util/resampling.py:111-118)."""
    periods = np.diff(st)
    err, written = 0, 0
    for i in range(len(sp) - 1):
        inerr = periods[i] * np.mean(sp[i:i + 2]) + err
        ni = int(round(inerr))
        err = inerr - ni
        written += ni
    return min(written, buflen)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    fourier, resampling, filters, wow, correlation = import_reference(a.ref)
    os.makedirs(a.out, exist_ok=True)

    def save(name, **arrs):
        path = os.path.join(a.out, name + ".npz")
        np.savez_compressed(path, **arrs)
        print(f"{name:28s} {os.path.getsize(path) / 1024:8.1f} KiB")

    # ------------------------------------------------------------------ STFT
    stft_cases = {
        # name: (n, seed, n_fft, hop, window, zeropad)
        "kat1": (4096, 0, 1024, 256, "hann", 1),
        "bh": (4096, 0, 1024, 256, "blackmanharris", 1),
        "small_vec": (1500, 1, 512, 32, "blackmanharris", 1),
        "odd_len": (3001, 2, 256, 64, "hann", 1),
        "zp2": (2000, 4, 256, 64, "hann", 2),
        "zp2_big": (5000, 4, 1024, 256, "blackmanharris", 2),
        "n2048": (5000, 5, 2048, 512, "hann", 1),
        "n64": (500, 6, 64, 16, "hamming", 1),
        "n4096": (9001, 7, 4096, 1024, "blackmanharris", 1),
        "hop_eq": (3000, 8, 128, 128, "hann", 1),
        "hop_odd": (2777, 9, 512, 100, "hann", 1),
    }
    out = {}
    for name, (n, seed, n_fft, hop, win, zp) in stft_cases.items():
        x = inputs.noise(n, seed)
        S = fourier.stft(x, n_fft, hop, win, zp)
        out[name + "_cfg"] = np.array([n, seed, n_fft, hop, zp], dtype=np.int64)
        out[name + "_win"] = np.array(win)
        out[name + "_S"] = np.asarray(S).astype(np.complex64)
        out[name + "_sum_abs"] = np.array(np.abs(S).sum())
        out[name + "_insum"] = np.array(inputs.checksum(x))
    # strided channel view, as stft receives it from (n, ch) arrays
    st = np.stack((inputs.noise(3000, 10), inputs.noise(3000, 11)), axis=-1)
    out["strided_S"] = np.asarray(fourier.stft(st[:, 1], 512, 128, "blackmanharris", 1)).astype(np.complex64)
    m = fourier.get_mag(inputs.noise(4096, 0), 1024, 256, "blackmanharris", 1)
    out["kat2_mag"] = np.asarray(m).astype(np.float64)
    save("stft", **out)

    # ----------------------------------------------------------------- ISTFT
    out = {}
    for name, (n, seed, n_fft, hop) in {"rt512": (6000, 12, 512, 32), "rt1024": (9000, 13, 1024, 256),
                                         "rt256": (4000, 14, 256, 64)}.items():
        x = inputs.noise(n, seed)
        S = np.array(fourier.stft(x, n_fft, hop))            # blackmanharris default
        S2 = S.copy()
        S2[5:40, 3:9] *= 0.25                                 # a modified spectrogram
        y = fourier.istft(S.copy(), hop_length=hop, length=n)
        y2 = fourier.istft(S2.copy(), hop_length=hop, length=n)
        y3 = fourier.istft(S.copy(), hop_length=hop)          # length=None trim branch
        out[name + "_cfg"] = np.array([n, seed, n_fft, hop], dtype=np.int64)
        out[name + "_y"], out[name + "_ymod"], out[name + "_ynolen"] = y, y2, y3
    # dropout-healer framing: fix_length(n + fft/2) first (dropout_healer_gui.py:127-133,164)
    x = inputs.noise(5000, 15)
    xp = fourier.fix_length(x, len(x) + 256)
    S = np.array(fourier.stft(xp, n_fft=512, step=32))
    out["heal_y"] = fourier.istft(S.copy(), length=len(x), hop_length=32)
    out["heal_shape"] = np.array(S.shape)
    save("istft", **out)

    # ----------------------------------------------------------- speed_to_pos
    out = {}
    n = 8192
    st = np.linspace(0, n, 33)
    sp = 1 + 0.01 * np.sin(2 * np.pi * np.arange(33) / 16 + 0.7)
    out["kat3_pos"] = resampling.speed_to_pos(st, sp, n)
    # test_sinc's own ramp (util/resampling.py:270-273), scaled down
    out["ramp_pos"] = resampling.speed_to_pos(np.array((0.0, 20000.0)), np.array((0.5, 2.0)), 20000)
    # bench-shaped curve, 2 s at 48 kHz
    sc = inputs.bench_speed_curve(2.0, 48000)
    out["bench_pos"] = resampling.speed_to_pos(sc[:, 0] * 48000, sc[:, 1], 96000)
    # fast wobble + noise on the speeds, hop 64
    rng = np.random.default_rng(21)
    st = np.linspace(0, 30000, 30000 // 64)
    sp = 1 + 0.05 * np.sin(np.arange(len(st)) * 0.3) + 0.002 * rng.standard_normal(len(st))
    out["wobble_st"], out["wobble_sp"] = st, sp
    out["wobble_pos"] = resampling.speed_to_pos(st, sp, 30000)
    # curve that starts late and stops short of the signal end: no trim -> written prefix only
    st = np.linspace(100.0, 7000.0, 28)
    sp = 1 + 0.02 * np.cos(np.arange(28) * 0.5)
    full = resampling.speed_to_pos(st, sp, 10000)
    out["untrimmed_st"], out["untrimmed_sp"] = st, sp
    out["untrimmed_pos"] = full[:written_len(st, sp, len(full))]
    out["untrimmed_buflen"] = np.array(len(full))
    save("speed_to_pos", **out)

    # ------------------------------------------------------------------- sinc
    out = {}
    sig = inputs.sine(8192, 440, 44100)
    out["kat4_y"] = resampling.sinc_wrapper(out_pos := np.load(os.path.join(a.out, "speed_to_pos.npz"))["kat3_pos"], sig, 0, 32)
    # identity positions, NT=8: exposes the leading-edge quirk at indices 0..7
    sig = inputs.noise(600, 30)
    out["ident_y"] = resampling.sinc_wrapper(np.arange(600, dtype=np.float64), sig, 0, 8)
    # ramp 0.5 -> 2 with NT=50 (GUI default quality): fc == 1 and fc < 1 regimes
    sig = (inputs.sine(20000, 440, 44100, 0.5) + inputs.sine(20000, 21000, 44100, 0.1)).astype(np.float32)
    rp = np.load(os.path.join(a.out, "speed_to_pos.npz"))["ramp_pos"]
    out["ramp_y"] = resampling.sinc_wrapper(rp, sig, 0, 50)
    # bench-shaped: hash-noise signal, +-1 % curve, NT=32
    sig = inputs.bench_signal(0, 96000, 48000)
    bp = np.load(os.path.join(a.out, "speed_to_pos.npz"))["bench_pos"]
    out["bench_y"] = resampling.sinc_wrapper(bp, sig, 0, 32)
    # positions running past the end of the signal (trailing clip, then empty window -> 0)
    sig = inputs.noise(1000, 31)
    pos = np.linspace(900.25, 1100.0, 300)
    out["tail_pos"] = pos
    out["tail_y"] = resampling.sinc_wrapper(pos, sig, 0, 16)
    # tiny NT and big NT
    sig = inputs.noise(3000, 32)
    pos = np.cumsum(np.full(2500, 1.013)) - 0.4
    out["nt1_y"] = resampling.sinc_wrapper(pos, sig, 0, 1)
    out["nt100_y"] = resampling.sinc_wrapper(pos, sig, 0, 100)
    # strong down-sampling (fc ~ 0.4) and repeated positions (period_to floor 1e-12)
    pos = np.concatenate((np.arange(100, 2900, 2.5), [1500.0, 1500.0, 1500.0, 1501.0]))
    out["down_pos"] = pos
    out["down_y"] = resampling.sinc_wrapper(pos, sig, 0, 24)
    # MT wrapper invariant the reference documents (util/resampling.py:277): mt == single
    buf = np.zeros((len(bp), 2), dtype=np.float32)
    resampling.sinc_wrapper_mt(buf[:, 1], bp, inputs.bench_signal(0, 96000, 48000), 0, 32)
    out["bench_y_mt"] = buf[:, 1].copy()
    out["mt_threads"] = np.array(os.cpu_count())
    save("sinc", **out)

    # ---------------------------------------------------------------- filters
    out = {}
    x = inputs.noise(4096, 0).astype(np.float64)
    out["band"] = filters.butter_bandpass_filter(x, 1000, 4000, 44100, order=3)
    out["low"] = filters.butter_bandpass_filter(x, 0, 20, 172.265625, order=3)
    out["high"] = filters.butter_bandpass_filter(x, 300, 0, 44100, order=3)
    out["band5"] = filters.butter_bandpass_filter(x, 500, 2000, 44100)
    out["pass_is_identity"] = np.array(filters.butter_bandpass_filter(x, 0, 0, 44100) is x)
    out["mavg"] = filters.moving_average(x[:100], 5)
    save("filters", **out)

    # ------------------------------------------------------------ correlation
    out = {}
    out["parabolic"] = np.array(correlation.parabolic([1, 3, 2], 1))
    aa = np.sin(np.arange(521) * 1.0)
    bb = np.sin(np.arange(521) * 1.0 + 3)
    out["find_delay"] = np.array(correlation.find_delay(aa.copy(), bb.copy(), window_name="hann"))
    out["xcorr_same"] = correlation.xcorr(inputs.noise(200, 40).astype(np.float64),
                                          inputs.noise(200, 41).astype(np.float64), mode="same")
    save("correlation", **out)

    # --------------------------------------------------------------- trackers
    out = {}
    sr, n, n_fft, hop = 48000, 72000, 1024, 256
    x = inputs.pilot(n, sr)
    spec = fourier.get_mag(x, n_fft, hop, "blackmanharris", 1)
    sig2d = x[:, None]
    trail = [(0.2, 4000.0), (1.3, 4000.0)]
    for name in ("Peak", "Peak Track", "Center of Gravity", "Zero-Crossing", "Correlation", "Freehand Draw"):
        tr = wow.wow_detectors[name](spec, sig2d, list(trail), n_fft, hop, sr, 0.5, "Linear")
        key = name.lower().replace(" ", "_").replace("-", "_")
        out[key + "_times"], out[key + "_freqs"] = tr.times, tr.freqs
    # a sloped, unsorted, 3-point trail with wide tolerance
    trail2 = [(1.2, 4030.0), (0.1, 3980.0), (0.6, 4010.0)]
    tr = wow.wow_detectors["Peak"](spec, sig2d, list(trail2), n_fft, hop, sr, 2.0, "Linear")
    out["peak2_times"], out["peak2_freqs"] = tr.times, tr.freqs
    out["cfg"] = np.array([sr, n, n_fft, hop])
    save("trackers", **out)

    # ------------------------------------- P0: config-3 data flow on L2 calls
    # glue restated from pyrespeeder_gui.py:119-140,165-191 and util/markers.py:182-226,585-639
    tr = wow.wow_detectors["Peak"](spec, sig2d, [(0.05, 4000.0), (1.45, 4000.0)], n_fft, hop, sr, 0.5, "Linear")
    log2speed = np.log2(tr.freqs)
    log2speed -= np.mean(log2speed)
    duration = n / sr
    marker_sr = sr / hop
    times = np.linspace(0, duration, num=int(duration * marker_sr))
    col = np.zeros((len(times), 1), dtype=np.float32)
    col[:, 0] = np.interp(times, tr.times, log2speed, left=np.nan, right=np.nan)
    mean = np.nanmean(col, axis=1)
    wow.interp_nans(mean)
    filt = filters.butter_bandpass_filter(mean, 0, 20, marker_sr, order=3)
    curve = np.stack((times, filt), axis=-1)
    np.power(2, curve[:, 1], curve[:, 1])
    pos = resampling.speed_to_pos(curve[:, 0] * sr, curve[:, 1], n)
    pos = pos[:written_len(curve[:, 0] * sr, curve[:, 1], len(pos))]
    y = resampling.sinc_wrapper(pos, x, 0, 32)
    save("pipeline", track_times=tr.times, track_freqs=tr.freqs, curve=curve, pos=pos, y=y,
         cfg=np.array([sr, n, n_fft, hop]))
    PIPE = dict(tr=tr, curve=curve, sr=sr, hop=hop, duration=duration)

    # ------------------------------------- config 4: the dropout healer's OWN glue (r04)
    # dropout_healer_gui.Canvas.resample_files (:111-166) and the batch-detection branch of .on_mouse_release (:168-242) are
    # called themselves, as plain functions on a stand-in canvas, with stand-ins for PyQt5 / vispy (oracle/ref_gui.py).  Until
    # r03 these fixtures came from oracle_np's restatements of the two methods; the arrays did not change by a bit.
    sys.path.insert(0, ROOT)
    from oracle import oracle_np, ref_gui
    D, P, M, W = ref_gui.import_reference_gui(a.ref)
    sr_h = 44100
    xh = (inputs.sine(30000, 1500.0, sr_h, 0.4) + 0.05 * inputs.noise(30000, 60)).astype(np.float32)
    xh[9000:9300] *= 0.05                                   # two synthetic dropouts
    xh[20000:20500] *= 0.1
    marks = [(0.2000, 500.0, 0.2110, 6000.0, 0.5), (0.4500, 800.0, 0.4680, 9000.0, 0.5),
             (0.2050, 1000.0, 0.2150, 3000.0, 1.0)]      # the third overlaps the first (gain clip path)
    healed, suffix_h = ref_gui.heal_through_reference(D, M, xh[:, None], sr_h, marks, 512, 32)
    assert suffix_h == "_drops"
    restated = oracle_np.heal_dropouts(xh, sr_h, marks, 512, 32,
                                       stft_fn=lambda x, n_fft, step: fourier.stft(x, n_fft=n_fft, step=step),
                                       istft_fn=lambda S, length, hop_length: fourier.istft(S, length=length, hop_length=hop_length))
    assert np.array_equal(healed, restated.astype(healed.dtype)), "oracle_np.heal_dropouts no longer restates the reference's glue"
    save("heal", y=healed[:, 0].astype(np.float64), marks=np.array(marks), sr=np.array(sr_h))

    xd = inputs.detect_input(sr_h)
    md = fourier.get_mag(xd, 512, 32, "blackmanharris", 1)
    det_args = (sr_h, 512, 32, 0.1, 2.6, 3000, 12000)
    vol_d, fb_d = oracle_np.band_volume_db(md, *det_args)
    found = ref_gui.detect_through_reference(D, M, md, sr_h, 512, 32, 0.1, 2.6, 3000, 12000, width_ms=20, sensitivity=5)
    found_np = oracle_np.detect_dropouts(md, *det_args, width_ms=20, sensitivity=5)
    assert np.array_equal(np.array(found), np.array(found_np)), "oracle_np.detect_dropouts no longer restates the reference's glue"
    save("detect", vol=vol_d, frame_b=np.array(fb_d), args=np.array(det_args, dtype=np.float64),
         found=np.array([(a_[0], a_[1], b_[0], b_[1]) for a_, b_ in found]))

    # ------------------------------ configs 1 and 3 on the reference's own sample files
    # The FLAC files are DATA copied into tests/golden/ (the reference has no tests; these are its demo
    # inputs, named by BASELINE.json configs 1 and 3).  soundfile is absent, so they are decoded with the
    # build's own decoder, which verifies the STREAMINFO MD5 of the decoded PCM.
    from pyaudiorestoration_amd import io_ops as my_io
    def grid(a, k=97):
        return np.asarray(a).ravel()[::k].copy()
    x1, sr1, _ = my_io.read_flac(os.path.join(a.ref, "samples", "flutter.flac"))
    m1 = fourier.get_mag(x1[:, 0], 1024, 256, "hann", 1)
    x3, sr3, _ = my_io.read_flac(os.path.join(a.ref, "samples", "flutter_192.flac"))
    spec3 = fourier.get_mag(x3[:, 0], 1024, 256, "blackmanharris", 1)
    tr3 = wow.wow_detectors["Peak"](spec3, x3, [(0.2, 4000.0), (4.0, 4000.0)], 1024, 256, sr3, 0.5, "Linear")
    l2s = np.log2(tr3.freqs)
    l2s -= np.mean(l2s)
    dur3 = len(x3) / sr3
    msr = sr3 / 256
    times3 = np.linspace(0, dur3, num=int(dur3 * msr))
    col = np.zeros((len(times3), 1), dtype=np.float32)
    col[:, 0] = np.interp(times3, tr3.times, l2s, left=np.nan, right=np.nan)
    mean3 = np.nanmean(col, axis=1)
    wow.interp_nans(mean3)
    filt3 = filters.butter_bandpass_filter(mean3, 0, 20, msr, order=3)
    curve3 = np.stack((times3, filt3), axis=-1)
    np.power(2, curve3[:, 1], curve3[:, 1])
    pos3 = resampling.speed_to_pos(curve3[:, 0] * sr3, curve3[:, 1], len(x3))
    pos3 = pos3[:written_len(curve3[:, 0] * sr3, curve3[:, 1], len(pos3))]
    sel = np.concatenate((np.arange(0, 3000), np.arange(400000, 403000), np.arange(len(pos3) - 3000, len(pos3))))
    y3_sel = np.concatenate([resampling.sinc_wrapper(pos3[a0:a0 + 3001], x3[:, 0], 0, 32)[:3000]
                             for a0 in (0, 400000)] +
                            [resampling.sinc_wrapper(pos3[len(pos3) - 3000:], x3[:, 0], 0, 32)])
    # the WHOLE config-3 output of the reference, every 8th sample kept (r05: the end-to-end test counts the samples beyond
    # the 1e-5 contract instead of bounding the max; 101 k samples = 400 KB)
    y3_full = resampling.sinc_wrapper(pos3, x3[:, 0], 0, 32)
    y3_dense = y3_full[5::8].copy()
    save("samples", c3_y_dense=y3_dense, c3_y_peak=np.array(np.max(np.abs(y3_full))), c1_shape=np.array(m1.shape), c1_sum=np.array(m1.sum()), c1_grid=grid(m1), c1_sr=np.array(sr1),
         c1_n=np.array(len(x1)), c3_shape=np.array(spec3.shape), c3_sum=np.array(spec3.sum()), c3_grid=grid(spec3),
         c3_sr=np.array(sr3), c3_n=np.array(len(x3)), c3_track_times=tr3.times, c3_track_freqs=tr3.freqs, c3_curve=curve3,
         c3_len_pos=np.array(len(pos3)), c3_pos_grid=grid(pos3, 1009), c3_sel=sel, c3_y_sel=y3_sel)

    # ------------------------------ heuristic dropout repair: dropouts_gui.MainWindow.process_heuristic itself (r05, 8f-3)
    hs = inputs.heuristic_input()
    h_low = ref_gui.heuristic_through_reference(a.ref, hs, 8000, 512, 64, max_width=0.06, f_upper=120, f_lower=20)
    xd, sr_d, _ = my_io.read_flac(os.path.join(a.ref, "samples", "dropouts_sample.flac"))
    h_gui = ref_gui.heuristic_through_reference(a.ref, xd, sr_d, 512, 64)          # the GUI's defaults (util/widgets.py:832-889)
    save("heuristic", low=h_low, low_params=np.array([8000, 512, 64, 0.06, 0.5, 3, 2, 120, 20]), low_in_sum=np.array(inputs.checksum(hs.ravel())),
         gui_every3=h_gui[::3].copy(), gui_peak=np.array(np.max(np.abs(h_gui))), gui_changed=np.array(np.max(np.abs(h_gui - xd))),
         gui_params=np.array([sr_d, 512, 64, 0.02, 0.5, 3, 2, 12000, 3000]))

    # ------------------------------ the master curves through the reference's own marker classes, and two .spd projects (r04)
    # MasterSpeedLine.update / MasterRegLine.update / Canvas.get_speed_curve (util/markers.py:625-708, pyrespeeder_gui.py:133-
    # 140) run themselves on TraceLine / RegLine objects built by from_cfg, as util/widgets.py:1247-1262 does for a .spd file.
    c_pilot, _, _ = ref_gui.speed_curve_through_reference(P, M, PIPE["sr"], PIPE["hop"], PIPE["duration"],
                                                           [[list(PIPE["tr"].times), list(PIPE["tr"].freqs), 0]], [])
    assert np.array_equal(c_pilot, PIPE["curve"]), "the pilot's master curve is not MasterSpeedLine.update's"
    assert np.array_equal(ref_gui.speed_curve_through_reference(P, M, sr3, 256, dur3, [[list(tr3.times), list(tr3.freqs), 0]], [])[0],
                          curve3), "config 3's master curve is not MasterSpeedLine.update's"
    # two overlapping traces of the 4 kHz pilot (the second one an octave-locked 0.02 up), then a sine regression over them
    trA = wow.wow_detectors["Peak"](spec3, x3, [(0.2, 4000.0), (2.4, 4000.0)], 1024, 256, sr3, 0.5, "Linear")
    trB = wow.wow_detectors["Peak Track"](spec3, x3, [(1.9, 4000.0), (4.0, 4000.0)], 1024, 256, sr3, 0.5, "Linear")
    lines = [[list(map(float, trA.times)), list(map(float, trA.freqs)), 0.0], [list(map(float, trB.times)), list(map(float, trB.freqs)), 0.02]]
    curve_t, ms_t, _ = ref_gui.speed_curve_through_reference(P, M, sr3, 256, dur3, lines, [], bands=(0.0, 20.0))
    amp, omega, phase, off = W.trace_sine_reg(curve_t, 0.5, 3.5, None)
    regs = [[0.5, 3.5, float(amp), float(omega), float(phase), float(off)]]
    curve_r, _, mr_r = ref_gui.speed_curve_through_reference(P, M, sr3, 256, dur3, lines, regs, bands=(0.0, 20.0))
    assert not np.array_equal(curve_r, curve_t)
    from util.config import save_json
    common = {"fft_size": 1024, "fft_overlap": 4, "fft_zeropad": 1, "mode": "Peak", "tolerance": 0.5, "highpass": 0.0, "lowpass": 20.0,
              "suffix": "", "sinc_quality": 32, "resampling_mode": "Sinc", "source": "flutter_192.flac"}
    save_json(os.path.join(a.out, "flutter_192_traces.spd"), dict(common, lines=lines, regs=[]))
    save_json(os.path.join(a.out, "flutter_192_reg.spd"), dict(common, lines=lines, regs=regs))
    spd = {}
    for tag, cv in (("traces", curve_t), ("reg", curve_r)):
        ps = resampling.speed_to_pos(cv[:, 0] * sr3, cv[:, 1], len(x3))
        ps = ps[:written_len(cv[:, 0] * sr3, cv[:, 1], len(ps))]
        starts = (0, 300000, len(ps) - 3001)
        spd[tag + "_curve"] = cv
        spd[tag + "_len_pos"] = np.array(len(ps))
        spd[tag + "_pos_grid"] = grid(ps, 1009)
        spd[tag + "_sel"] = np.concatenate([np.arange(a0, a0 + 3000) for a0 in starts])
        spd[tag + "_y_sel"] = np.concatenate([resampling.sinc_wrapper(ps[a0:a0 + 3001], x3[:, 0], 0, 32)[:3000] for a0 in starts])
    save("spd", reg=np.array(regs[0]), master_traces=ms_t, master_reg=mr_r, **spd)

    # ----------------------------------------------- Linear mode + lag curve
    sig = inputs.noise(5000, 50)
    lag = np.array([[0.0, 0.0], [0.02, 0.0005], [0.06, -0.001], [0.1, 0.002]])
    srl = 48000
    st_, lg_ = lag[:, 0] * srl, lag[:, 1] * srl
    num_out = len(sig) + abs(lg_[-1])
    sa = np.interp(np.arange(num_out), st_, st_ - lg_)
    cut = resampling.find_cutoff(sa, len(sig))
    if cut is not None:
        sa = sa[:cut[0]]
    np.clip(sa, 0, None, out=sa)
    lin = np.interp(sa, np.arange(len(sig)), sig, left=0.0, right=0.0)
    save("linear_lag", lag=lag, pos=sa, lin=lin.astype(np.float32), sr=np.array(srl))


if __name__ == "__main__":
    main()
