"""P0 -- headless restatement of the pyrespeeder data flow (BASELINE config 3).

The reference spreads these ~30 lines of array math over GUI classes that cannot run headless:
pyrespeeder_gui.py:119-140 (run_resample / get_speed_curve), :165-191 (tracker invocation),
util/markers.py:182-226 (TraceLine: log2 speed centred on 0), :585-639 (BaseLine.get_times,
marker_sr, sample_lines, filter_bandpass, MasterSpeedLine.update, get_linspace) and
util/spectrum.py:384-385 (spectra are computed with 'blackmanharris').
Here: STFT magnitude (K_stft) -> tracker (K_track) -> master speed curve (K_sosfiltfilt) ->
positions (K_pos) -> sinc resample (K_sinc), with the signal and spectrogram resident in HBM.
"""
import warnings

import logging

import numpy as np
import scipy.signal
import torch

from . import _dev, filters, fourier, resampling, wow_detection


def trace_to_speed(freqs):
    """TraceLine.__init__ (util/markers.py:197-199): log2 speed, centred on 0 (offset 0)."""
    speed = np.log2(freqs)
    return speed - np.mean(speed)


def master_speed_curve(lines, duration, sr, hop, bands=(0, 20)):
    """MasterSpeedLine.update + get_linspace (util/markers.py:585-639).
    lines: list of (times, log2_speed).  Returns [[t_seconds, linear_speed], ...]."""
    marker_sr = sr / hop
    times = np.linspace(0, duration, num=int(duration * marker_sr))
    out = np.zeros((len(times), len(lines)), dtype=np.float32)
    for i, (line_times, line_values) in enumerate(lines):
        out[:, i] = np.interp(times, line_times, line_values, left=np.nan, right=np.nan)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        mean_with_nans = np.nanmean(out, axis=1)
    wow_detection.interp_nans(mean_with_nans)
    lowcut, highcut = sorted(bands)
    filtered = filters.butter_bandpass_filter(mean_with_nans, lowcut, highcut, marker_sr, order=3)
    data = np.stack((times, filtered), axis=-1)
    curve = np.array(data)
    np.power(2, curve[:, 1], curve[:, 1])
    return curve


def master_reg_curve(regs, duration, sr, hop):
    """MasterRegLine.update + get_linspace (util/markers.py:670-708, 593-597): the sine regressions, sorted by their centres,
    joined by interpolating phase and amplitude between the centres.  regs: rows (t0, t1, amplitude, omega, phase, offset) as
    RegLine.to_cfg writes them (util/markers.py:175-176).  Returns [[t_seconds, linear_speed], ...]."""
    marker_sr = sr / hop
    times = np.linspace(0, duration, num=int(duration * marker_sr))
    if len(times) == 0:                              # a file shorter than one hop: the reference indexes times[0] (IndexError)
        raise ValueError(f"file too short for a speed curve: duration {duration} s at {marker_sr} markers/s")
    rows = []
    for t0, t1, amplitude, omega, phase, offset in regs:
        if amplitude < 0:                            # RegLine.__init__ (:117-119): a negative amplitude is a phase of pi
            amplitude, phase = -amplitude, phase + np.pi
        rows.append(((t0 + t1) / 2, amplitude, omega, phase, offset))
    rows.sort(key=lambda r: r[0])
    pi2 = 2 * np.pi
    t_centers, amp_centers, phi_centers = [], [], []
    for i, (tc, amplitude, omega, phase, offset) in enumerate(rows):
        if i == 0:
            phi_centers.append(omega * times[0] + phase % pi2 + offset * pi2)
            t_centers.append(times[0])
            amp_centers.append(amplitude)
        phi_centers.append(omega * tc + phase % pi2 + offset * pi2)
        t_centers.append(tc)
        amp_centers.append(amplitude)
        if i == len(rows) - 1:
            phi_centers.append(omega * times[-1] + phase % pi2 + offset * pi2)
            t_centers.append(times[-1])
            amp_centers.append(amplitude)
    sine_curve = np.sin(np.interp(times, t_centers, phi_centers))
    amplitudes_sampled = np.interp(times, t_centers, amp_centers)
    data = np.stack((times, 1.5 * amplitudes_sampled * sine_curve), axis=-1)       # (:704: "boost it a bit")
    np.power(2, data[:, 1], data[:, 1])
    return data


def project_speed_curve(cfg, duration, sr):
    """The speed curve a saved pyrespeeder project (.spd: pyrespeeder_gui.py:17-18 STORE = lines, regs; util/widgets.py:1224-
    1262) resamples with: Canvas.get_speed_curve (pyrespeeder_gui.py:133-140) -- the regressed curve if the project holds
    regressions, else the measured one (the traces' mean, band-passed between the project's highpass and lowpass).
    cfg["lines"]: rows (times, freqs, offset) of TraceLine.to_cfg (util/markers.py:275-276)."""
    hop = cfg["fft_size"] // cfg.get("fft_overlap", 1)
    if cfg.get("regs"):
        return master_reg_curve(cfg["regs"], duration, sr, hop)
    lines = []
    for times, freqs, offset in cfg.get("lines", ()):
        lines.append((np.asarray(times, dtype=np.float64), trace_to_speed(np.asarray(freqs, dtype=np.float64)) + (offset or 0)))
    if not lines:
        raise ValueError("project holds neither traces nor regressions")
    return master_speed_curve(lines, duration, sr, hop, (cfg.get("highpass", 0), cfg.get("lowpass", 20)))


def respeed_project(project, source=None, out_suffix=None, device=None, sinc_quality=None, resampling_mode=None):
    """Run a saved pyrespeeder project headless: `project` is the path of a .spd JSON (fft_size, fft_overlap,
    highpass, lowpass, lines, regs, source, resampling_mode, sinc_quality, suffix) or the dict itself; `source` overrides
    the audio path stored in it; out_suffix / sinc_quality / resampling_mode override the project's own settings (None: the
    project's).  Writes <source>_res<suffix>.wav through resampling.run like Canvas.run_resample (pyrespeeder_gui.py:119-131)
    and returns the speed curve."""
    import json
    from . import io_ops
    if isinstance(project, (str, bytes)) or hasattr(project, "__fspath__"):
        with open(project) as fh:
            cfg = json.load(fh)
    else:
        cfg = dict(project)
    path = source or cfg["source"]
    signal, sr, _ = io_ops.read_file(path)
    curve = project_speed_curve(cfg, len(signal) / sr, sr)
    # resampling.run works on the CURRENT device: make `device` that one for the duration of the call (the argument
    # was accepted and ignored, which only worked because the CLI's worker thread had called set_device itself)
    with torch.cuda.device(_dev.device_index(device)):
        resampling.run((path,), signal_data=((signal, sr),), speed_curve=curve,
                       resampling_mode=resampling_mode or cfg.get("resampling_mode", "Sinc"),
                       sinc_quality=sinc_quality or cfg.get("sinc_quality", 50),
                       suffix=cfg.get("suffix", "") if out_suffix is None else out_suffix)
    return curve


def respeed(signal, sr, trail, fft_size=1024, hop=256, zeropad=1, mode="Peak", tolerance_st=0.5, bands=(0, 20),
            sinc_quality=32, resampling_mode="Sinc", device=None):
    """signal: float32 (n,) or (n, ch).  Returns dict with every intermediate the reference's GUI
    would hold: spectrum (device), track times/freqs, speed curve, positions (device), output."""
    dev = _dev.device_index(device)
    sig2d = signal[:, None] if signal.ndim == 1 else signal
    n = sig2d.shape[0]
    sig_t = _dev.to_dev(sig2d, torch.float32, dev)                       # (n, ch) resident in HBM
    ch = sig2d.shape[1]
    chan0 = sig_t.reshape(-1)[0::ch] if ch > 1 else sig_t.reshape(-1)
    spec = fourier.get_mag(chan0, fft_size, hop, "blackmanharris", zeropad)      # device tensor (bins, frames)
    # Peak / Peak Track read their band from the signal in float64 (what the reference's numpy backend hands them)
    refine = {"x": chan0, "n_fft": fft_size, "zeropad": zeropad,
              "window": _dev.to_dev(scipy.signal.get_window("blackmanharris", fft_size).astype(np.float32), torch.float32, dev)}
    track = wow_detection.wow_detectors[mode](spec, sig2d, list(trail), fft_size * zeropad, hop, sr, tolerance_st,
                                              "Linear", refine=refine)
    curve = master_speed_curve([(track.times, trace_to_speed(track.freqs))], n / sr, sr, hop, bands)
    st_t = _dev.to_dev(curve[:, 0] * sr, torch.float64, dev)
    sp_t = _dev.to_dev(np.ascontiguousarray(curve[:, 1]), torch.float64, dev)
    pos_t = resampling.speed_to_pos_dev(st_t, sp_t, n, dev)      # kept: the GUI shows / reuses sample_at
    plan = resampling.speed_plan_dev(st_t, sp_t, n, dev, fused=True) if resampling_mode == "Sinc" else None
    out_t = _dev.empty((pos_t.numel(), ch), torch.float32, dev)
    flat_in, flat_out = sig_t.reshape(-1), out_t.reshape(-1)
    layout = dict(sig_stride=ch, len_in=n, out_stride=ch)
    c = 0
    while c < ch:
        if resampling_mode == "Sinc" and plan.fused_ok and c + 1 < ch:      # channel pairs share one stereo launch
            resampling.varispeed_fused_stereo_dev(plan, flat_in[c:], flat_in[c + 1:], sinc_quality, flat_out[c:],
                                                  flat_out[c + 1:], **layout)
            c += 2
            continue
        if resampling_mode == "Sinc" and plan.fused_ok:
            resampling.varispeed_fused_dev(plan, flat_in[c:], sinc_quality, flat_out[c:], **layout)
        elif resampling_mode == "Sinc":
            resampling.sinc_resample_dev(pos_t, flat_in[c:], sinc_quality, flat_out[c:], dev=dev, **layout)
        else:
            resampling.linear_resample_dev(pos_t, flat_in[c:], flat_out[c:], dev=dev, **layout)
        c += 1
    return {"spectrum": spec, "times": track.times, "freqs": track.freqs, "speed_curve": curve, "positions": pos_t,
            "output": out_t}


# ---------------------------------------------------------------------- config 4: dropout healer
def to_dB(a):
    """util/units.py:24-25."""
    return 20 * np.log10(a)


def marker_geometry(marker, sr, hop, fft_size):
    """(frame_b, frame_a, frame_surrounding, bin_l, bin_u) of one DropoutSample.to_cfg() tuple
    (a0, a1, b0, b1, surrounding): the integer arithmetic of dropout_healer_gui.py:99-109,136-142 on
    width/t/f/height as util/markers.py:368-388 derives them from the two corners."""
    a0, a1, b0, b1, surrounding = marker

    def t2f(t):
        return int(t * sr / hop)

    def f2b(f):
        return max(1, min(fft_size // 2, int(round(f * fft_size / sr))))

    width, t = abs(a0 - b0), (a0 + b0) / 2
    f, height = (a1 + b1) / 2, abs(a1 - b1)
    return (t2f(t - width / 2), t2f(t + width / 2), max(1, t2f(width * surrounding)),
            f2b(f - height / 2), f2b(f + height / 2))


def _check_geometry(geo, frames):
    geo = np.asarray(geo, dtype=np.int64).reshape(-1, 5)
    if len(geo) == 0:
        return
    fb, fa, fs, bl, bu = (geo[:, k] for k in range(5))
    # the reference would slice with a negative start / past the end and average an empty slice (NaN
    # gain -> NaN audio), or hand RegularGridInterpolator an empty axis; refuse both
    bad = (fb - fs < 0) | (fa + fs > frames) | (fa - fb < 1)
    if np.any(bad):
        k = int(np.flatnonzero(bad)[0])
        raise ValueError(f"dropout marker frames [{fb[k]}-{fs[k]}, {fa[k]}+{fs[k]}) leave the {frames}-frame spectrogram or are empty")
    if np.any(bu - bl < 1):
        raise ValueError("dropout marker spans no frequency bin")


def heal_spectrum_dev(spec_fm, geometry, dev=None, gain=None):
    """In-place inpaint of a frame-major complex64 device spectrogram: mask for all markers (one launch), then
    apply-and-clear over the boxes only (one launch).  Returns the (all-zero again) mask for reuse."""
    from . import _lib
    dev = _dev.device_index(dev)
    frames, bins = spec_fm.shape
    if torch.is_tensor(geometry):                        # already checked and on the device: int32 [n][5]
        geo, geo_t = geometry, geometry
    else:
        geo = np.asarray(geometry, dtype=np.int64).reshape(-1, 5)
        _check_geometry(geo, frames)
        geo_t = None
    if gain is None:
        gain = torch.zeros((frames, bins), dtype=torch.float32, device=spec_fm.device)
    if len(geo):
        L = _lib.lib()
        if geo_t is None:
            geo_t = _dev.to_dev(np.ascontiguousarray(geo, dtype=np.int32), torch.int32, dev)
        args = (dev, _dev.ptr(spec_fm), frames, bins, _dev.ptr(geo_t), len(geo), _dev.ptr(gain), _dev.stream_ptr(dev))
        _lib.check(L.par_inpaint_gain_db_c64(*args))
        _lib.check(L.par_spec_apply_gain_boxes_c64(*args))
    return gain


def heal_segments(geometry, frames_total, n_pad, n, fft_size, hop):
    """Sparse plan of the healer: which frames can a marker influence, and how do they line up in one short pseudo-signal?
    A box modifies frames [fb, fa) and reads fs frames on either side; a modified frame reaches samples half a window away,
    and rebuilding those samples needs every frame that overlaps them: frames [min(fb - fs, fb - R + 1), max(fa + fs,
    fa + R - 1)) with R = fft_size / hop.  Overlapping ranges merge into segments [g0, g1); segment k becomes the samples
    [g0 hop - fft/2, (g1 - 1) hop + fft/2) of the padded signal, placed at a multiple of hop in the pseudo-signal so that
    its frames sit on the pseudo-signal's own frame grid.  A segment within R frames of either end of the file is extended
    to that end and keeps it as the pseudo-signal's end (the transform's reflect padding is then the original's).
    -> None when the dense path should be taken, else dict(src, dst, len: sample ranges to gather; frame_shift per box;
    v_src, v_dst, v_len: valid ranges to copy back; total)."""
    geo = np.asarray(geometry, dtype=np.int64).reshape(-1, 5)
    R = fft_size // hop
    if len(geo) == 0 or fft_size % (2 * hop) or R < 2:
        return None
    fb, fa, fs = geo[:, 0], geo[:, 1], geo[:, 2]
    lo = np.minimum(fb - fs, fb - R + 1)
    hi = np.maximum(fa + fs, fa + R - 1)
    order = np.argsort(lo, kind="stable")
    lo_s, hi_s = lo[order], hi[order]
    reach = np.maximum.accumulate(hi_s)                   # furthest frame any earlier box needs
    first = np.ones(len(geo), dtype=bool)
    first[1:] = lo_s[1:] > reach[:-1]                     # a box that starts behind everything before it opens a segment
    seg_id = np.cumsum(first) - 1
    seg_of = np.empty(len(geo), dtype=np.int64)
    seg_of[order] = seg_id
    g0 = lo_s[first]
    g1 = np.maximum.reduceat(hi_s, np.flatnonzero(first))
    n_seg = len(g0)
    half = fft_size // 2
    at_start, at_end = g0 < R, g1 > frames_total - R
    if np.any(at_start & at_end) or np.any(at_start[1:]) or np.any(at_end[:-1]):
        return None                                       # one segment spans the file (the others cannot happen when sorted)
    s0 = np.where(at_start, 0, g0 * hop - half)
    s1 = np.where(at_end, n_pad, (g1 - 1) * hop + half)
    v0 = np.where(at_start, 0, (g0 - 1) * hop + half)
    v1 = np.where(at_end, n, np.minimum(n, g1 * hop - half))
    ln = s1 - s0
    dst = np.concatenate(([0], np.cumsum(ln)[:-1]))
    at = int(np.sum(ln))
    shift = (dst - s0) // hop                             # pseudo-signal frame = original frame + shift
    keep = v1 > v0
    src, v_src, v_dst, v_len = s0, (dst + v0 - s0)[keep], v0[keep], (v1 - v0)[keep]
    segs = range(n_seg)
    if at > 0.6 * n_pad:
        return None
    return {"src": np.asarray(src, np.int64), "dst": np.asarray(dst, np.int64), "len": np.asarray(ln, np.int64),
            "frame_shift": np.asarray(shift, np.int64)[seg_of], "v_src": np.asarray(v_src, np.int64),
            "v_dst": np.asarray(v_dst, np.int64), "v_len": np.asarray(v_len, np.int64), "total": int(at), "segments": len(segs)}


def _copy_segments(src_t, src_stride, n_valid, n_padded, padded, src_start, dst_start, lens, dst_t, dst_stride, dev, cache=None,
                   key=None):
    """cache / key: a dict that keeps the device copy of the index arrays (a plan serves every channel of a file)."""
    from . import _lib
    if len(lens) == 0:
        return
    idx = cache.get((key, dev)) if cache is not None else None
    if idx is None:
        run = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int64)
        idx = _dev.to_dev(np.stack((src_start, dst_start, lens, run)), torch.int64, dev)
        if cache is not None:
            cache[(key, dev)] = idx
    _lib.check(_lib.lib().par_copy_segments_f32(dev, _dev.ptr(src_t), src_stride, n_valid, n_padded, int(padded), _dev.ptr(idx[0]),
                                                _dev.ptr(idx[1]), _dev.ptr(idx[2]), _dev.ptr(idx[3]), len(lens), int(np.sum(lens)),
                                                _dev.ptr(dst_t), dst_stride, _dev.stream_ptr(dev)))


def heal_dropouts_dev(sig_t, n, ch, c, geometry, fft_size, hop, out_t, dev, sparse=None, gain=None, plan=None):
    """One channel of the healer, device to device: sig_t (n, ch) float32 interleaved -> out_t (n, ch), channel c.
    sparse None: decide from the markers (heal_segments); True / False force a path (True still falls back to the dense
    path when the markers cover most of the file); plan: a heal_segments result for these markers (it depends on the
    markers only: one plan serves every channel of a file).  Returns (gain mask for reuse, plan or None)."""
    n_pad = n + fft_size // 2
    frames_total = n_pad // hop + 1
    if plan is None and sparse is not False:
        plan = heal_segments(geometry, frames_total, n_pad, n, fft_size, hop)
    flat_in, flat_out = sig_t.reshape(-1), out_t.reshape(-1)
    if plan is None:
        # the reference's own dataflow (dropout_healer_gui.py:120-164): pad, transform everything, heal, invert everything
        pad_t = _dev.empty(n_pad, torch.float32, dev)
        pad_t[:n] = flat_in[c::ch]
        pad_t[n:] = 0.0
        S = fourier.stft(pad_t, n_fft=fft_size, step=hop)                  # (bins, frames) device
        healed = S.T.contiguous()                                          # frame-major [frames][bins]
        gain = heal_spectrum_dev(healed, geometry, dev, gain)              # mask is zero again: reused by the next channel
        flat_out[c::ch] = fourier.istft(healed.T, length=n, hop_length=hop)
        return gain, None
    geo = np.array(geometry, dtype=np.int64).reshape(-1, 5)
    T = plan["total"]
    xs = _dev.empty(T, torch.float32, dev)
    cache = plan.setdefault("_device", {})
    _copy_segments(flat_in[c:], ch, n, n_pad, True, plan["src"], plan["dst"], plan["len"], xs, 1, dev, cache, "gather")
    S = fourier.stft(xs, n_fft=fft_size, step=hop)
    healed = S.T.contiguous()
    geo_t = cache.get(("geo", dev))
    if geo_t is None:
        geo[:, 0] += plan["frame_shift"]
        geo[:, 1] += plan["frame_shift"]
        _check_geometry(geo, healed.shape[0])
        geo_t = cache[("geo", dev)] = _dev.to_dev(np.ascontiguousarray(geo, dtype=np.int32), torch.int32, dev)
    if gain is not None and gain.shape != healed.shape:
        gain = None
    gain = heal_spectrum_dev(healed, geo_t, dev, gain)
    ys = fourier.istft(healed.T, length=T, hop_length=hop)
    if ch == 1 and flat_out.data_ptr() != flat_in.data_ptr():
        flat_out.copy_(flat_in)
    elif flat_out.data_ptr() != flat_in.data_ptr():
        flat_out[c::ch] = flat_in[c::ch]
    _copy_segments(ys, 1, T, T, False, plan["v_src"], plan["v_dst"], plan["v_len"], flat_out[c:], ch, dev, cache, "scatter")
    return gain, plan


def heal_dropouts(signal, sr, markers, fft_size=512, hop=32, channels=None, device=None, sparse=None):
    """Spectral inpainting of marked dropouts -- headless restatement of
    dropout_healer_gui.Canvas.resample_files (dropout_healer_gui.py:111-166).

    signal: float32 (n, ch).  markers: iterable of (a0, a1, b0, b1, surrounding) = the reference's
    DropoutSample.to_cfg() (util/markers.py:368-388, 424-426): corner (t, f) pairs and the
    surrounding factor.  STFT, the per-marker targets and gain mask (all markers in one launch), gain
    application and ISTFT run on the device; only the healed channel returns to the host.

    sparse (r03): the reference transforms the whole file, multiplies by a gain mask that is 1 outside the boxes and
    inverts the whole file; outside the reach of the boxes that round trip is the identity to 1.6e-8.  By default only
    the frames a box can influence are transformed (heal_segments) and the rest of the signal is copied; sparse=False
    takes the reference-shaped dense path."""
    dev = _dev.device_index(device)
    sig2d = signal[:, None] if signal.ndim == 1 else signal
    n, ch = sig2d.shape
    if channels is None:
        channels = range(ch)
    sig_t = _dev.to_dev(np.ascontiguousarray(sig2d), torch.float32, dev)    # (n, ch) in HBM
    out_t = _dev.empty((n, ch), torch.float32, dev)
    geometry = [marker_geometry(m, sr, hop, fft_size) for m in markers]
    _check_geometry(np.asarray(geometry, dtype=np.int64).reshape(-1, 5), (n + fft_size // 2) // hop + 1)
    gain, plan = None, None
    geometry = np.asarray(geometry, dtype=np.int64).reshape(-1, 5)
    for c in channels:
        gain, plan = heal_dropouts_dev(sig_t, n, ch, c, geometry, fft_size, hop, out_t, dev, sparse, gain, plan)
    out = np.empty(sig2d.shape, dtype=sig2d.dtype)                          # channels not selected stay uninitialised, like the reference's np.empty
    got = _dev.to_host(out_t)
    for c in channels:
        out[:, c] = got[:, c]
    return out


def band_volume_db(mag, sr, fft_size, hop, t_0, t_1, f_lower, f_upper, device=None):
    """Volume curve of the dropout detector (dropout_healer_gui.py:195-203): mean over bins
    [bin_l, bin_u) of to_dB(magnitude) for frames [frame_b, frame_a).  mag: (bins, frames) magnitude as
    fourier.get_mag returns it (device tensor, or numpy which is uploaded).  -> (vol f64 numpy, frame_b)."""
    from . import _lib
    dev = _dev.device_index(device)
    if not torch.is_tensor(mag):
        mag = _dev.to_dev(np.ascontiguousarray(np.asarray(mag, dtype=np.float32).T), torch.float32, dev).T
    fm = mag.T
    if not (fm.stride(1) == 1 and fm.stride(0) >= fm.shape[1]):
        fm = fm.contiguous()
    frames, bins = fm.shape
    frame_b, frame_a = int(t_0 * sr / hop), int(t_1 * sr / hop)
    f2b = lambda f: max(1, min(fft_size // 2, int(round(f * fft_size / sr))))
    bin_l, bin_u = f2b(f_lower), f2b(f_upper)
    frame_a = min(frame_a, frames)                                        # numpy slicing clamps the end
    vol = _dev.empty(max(0, frame_a - frame_b), torch.float64, dev)
    _lib.check(_lib.lib().par_band_mean_db_f32(dev, _dev.ptr(fm), frames, bins, fm.stride(0), bin_l, bin_u, frame_b, frame_a,
                                               _dev.ptr(vol), _dev.stream_ptr(dev)))
    return vol.cpu().numpy(), frame_b


def detect_dropouts(mag, sr, fft_size, hop, t_0, t_1, f_lower, f_upper, width_ms=20, sensitivity=5, device=None):
    """Batch dropout detection -- headless restatement of the Alt-drag branch of
    dropout_healer_gui.Canvas.on_mouse_release (dropout_healer_gui.py:185-242).  The band volume curve
    (the only pass over the spectrogram) runs on the device; the valley search on that short curve uses the
    same scipy calls as the reference (savgol_filter, find_peaks prominence, polyfit refinement).
    Returns [((t_before, f_lower), (t_after, f_upper)), ...] = the corner pairs the reference hands to
    DropoutSample (:240).  width_ms / sensitivity are DropoutWidget.width / .sensitivity (util/widgets.py:715-727)."""
    import scipy.signal
    from scipy.signal import savgol_filter
    vol, frame_b = band_volume_db(mag, sr, fft_size, hop, t_0, t_1, f_lower, f_upper, device)
    t2f = lambda t: int(t * sr / hop)
    f2t = lambda f: f / sr * hop
    half_width = width_ms / 1000 / 2
    frames_half_width = t2f(half_width)
    vol_lt = savgol_filter(vol, frames_half_width * 12, 5)
    vol_st = savgol_filter(vol, frames_half_width, 5)
    peaks, _ = scipy.signal.find_peaks(-vol, prominence=10.0 - sensitivity, rel_height=0.5)
    found = []
    for f_peak in peaks:
        t_center = f2t(frame_b + f_peak)
        try:
            f_qw = t2f(half_width / 4)
            xp = np.arange(f_peak - f_qw, f_peak + f_qw)
            parabola = np.poly1d(np.polyfit(xp, vol_st[f_peak - f_qw:f_peak + f_qw], 2))
            f_hw = t2f(half_width)
            f_before, f_after = f_peak - f_hw, f_peak + f_hw
            fp = parabola(np.arange(f_before, f_after))
            f_intersection = scipy.signal.argrelmin(np.abs(fp - vol_lt[f_before:f_after]))[0]
            assert len(f_intersection) == 2
            half_width = f2t(f_intersection[1] - f_intersection[0])       # (sic) carried over to later peaks, :232
        except Exception:
            logging.exception(f"Could not refine width at peak {f_peak}")
        found.append(((t_center - half_width, f_lower), (t_center + half_width, f_upper)))
    return found


# ---------------------------------------------------------------------- heuristic dropout repair (dropouts_gui)
def heuristic_bands(f_lower, f_upper, num_bands):
    """Band edges of dropouts_gui.py:252 (numpy uint16 scalars, like the reference's)."""
    return np.logspace(np.log2(f_lower), np.log2(f_upper), num=num_bands, endpoint=True, base=2, dtype=np.uint16)


def heuristic_bins(f_lower_band, f_upper_band, fft_size, sr):
    """dropouts_gui.py:281-282, int(f * fft_size / sr) with f a numpy uint16 scalar: under numpy >= 2 (NEP 50) the product stays
    uint16 and wraps from f * fft_size = 65536 on, so at 44.1 kHz / 512 points the reference's bands sit in bins 0..1 whatever
    their frequencies.  Evaluated through numpy itself: this follows the installed numpy exactly as the reference does."""
    with np.errstate(over="ignore"):
        return int(f_lower_band * fft_size / sr), int(f_upper_band * fft_size / sr)


def heuristic_gain_curve(vol, d, max_slope):
    """Gain curve (dB) of one band from its volume curve: the reference's valley search (scipy find_peaks on -vol, prominence
    5 dB) and straight-line patches, dropouts_gui.py:287-311 -- O(frames) host work on a curve the device produced."""
    n_frames = len(vol)
    peaks, _ = scipy.signal.find_peaks(-vol, height=None, threshold=None, distance=None, prominence=5, wlen=None, rel_height=0.5,
                                       plateau_size=None)
    gain_curve = np.zeros(n_frames)
    for peak_i in peaks:
        if 2 * d < peak_i < n_frames - 2 * d - 1:
            left = np.mean(vol[peak_i - 2 * d:peak_i - d])
            right = np.mean(vol[peak_i + d:peak_i + 2 * d])
            if abs((left - right) / (2 * d)) < max_slope:
                gain_curve[peak_i - d:peak_i + d + 1] = np.interp(range(2 * d + 1), (0, 2 * d), (left, right)) - vol[peak_i - d:peak_i + d + 1]
    return gain_curve


def heal_heuristic(signal, sr, fft_size=512, hop=64, max_width=0.02, max_slope=0.5, num_bands=3, bottom_freedom=2, f_upper=12000,
                   f_lower=3000, device=None):
    """Headless restatement of dropouts_gui.MainWindow.process_heuristic (dropouts_gui.py:241-323; defaults = DropoutWidget's,
    util/widgets.py:832-889): signal (n, ch) float32 (numpy or device tensor) -> repaired (n, ch) float32 of the same kind.
    Per band, from the top one down: band volume per channel (K_heal, device), valley gains (host, O(frames)), then over ALL
    channels at once signal x (factor - 1) (par_curve_scale_f64), the zero-phase order-3 band-pass (K_sosfiltfilt, batched) and
    the add into the signal the next band starts from (par_accumulate_f64_f32) -- the bands are sequential through the signal in
    the reference (:314-321), the channels are not."""
    from . import _lib
    dev = _dev.device_index(device)
    was_tensor = torch.is_tensor(signal)
    sig2d = signal if signal.ndim == 2 else signal[:, None]
    n, ch = sig2d.shape
    sig_t = _dev.to_dev(sig2d, torch.float32, dev).clone() if was_tensor else _dev.to_dev(np.ascontiguousarray(sig2d), torch.float32, dev)
    sig_t = sig_t.contiguous()
    flat = sig_t.reshape(-1)
    L = _lib.lib()
    bands = heuristic_bands(f_lower, f_upper, num_bands)
    d = int(max_width / 1.5 * sr / hop)
    # the spectrograms come from the channels as they are BEFORE any band is added (the reference takes them once per channel)
    mags = []
    for c in range(ch):
        m = fourier.get_mag(flat[c::ch] if ch > 1 else flat, fft_size, hop, "hann", 1)
        fm = m.T
        if not (fm.stride(1) == 1 and fm.stride(0) >= fm.shape[1]):
            fm = fm.contiguous()
        mags.append(fm)
    frames, bins = mags[0].shape
    fac = np.ones((ch, frames)) * 1000
    vol_t = _dev.empty(frames, torch.float64, dev)
    out_t = _dev.empty((ch, n), torch.float64, dev)
    with np.errstate(all="ignore"):
        for f_lo, f_hi in reversed(list(zip(bands[:-1], bands[1:]))):
            bin_l, bin_u = heuristic_bins(f_lo, f_hi, fft_size, sr)
            for c in range(ch):
                lo, hi = max(0, min(bin_l, bins)), max(0, min(bin_u, bins))          # numpy slicing clamps
                if lo < hi:
                    _lib.check(L.par_band_mean_db_f32(dev, _dev.ptr(mags[c]), frames, bins, mags[c].stride(0), lo, hi, 0, frames,
                                                      _dev.ptr(vol_t), _dev.stream_ptr(dev)))
                    vol = vol_t.cpu().numpy()
                else:
                    vol = np.full(frames, np.nan)                                  # np.mean of an empty slice
                fac[c] = np.clip(np.power(10, heuristic_gain_curve(vol, d, max_slope) / 20), 1, fac[c] * bottom_freedom)
            if not np.any(fac != 1.0):
                continue                                                            # the band adds filter(0) = 0 to the signal
            fm1_t = _dev.to_dev(np.ascontiguousarray(fac - 1.0), torch.float64, dev)
            _lib.check(L.par_curve_scale_f64(dev, _dev.ptr(flat), ch, ch, n, _dev.ptr(fm1_t), frames, _dev.ptr(out_t), _dev.stream_ptr(dev)))
            # (a band without a cut-off inside (0, Nyquist) comes back unfiltered, like the reference's butter_bandpass_filter)
            y_t = filters.bandpass_batch_dev(out_t, f_lo, f_hi, sr, order=3, dev=dev)
            _lib.check(L.par_accumulate_f64_f32(dev, _dev.ptr(flat), ch, ch, n, _dev.ptr(y_t), _dev.stream_ptr(dev)))
    res = sig_t if signal.ndim == 2 else sig_t[:, 0]
    return res if was_tensor else _dev.to_host(res)


# ---------------------------------------------------------------------- tape synchronisation (pytapesynch)
def lag_curve_from_markers(markers, duration, sr, hop, smoothing=3, bands=(0, 9999999)):
    """Lag curve (N, 2) = (time s, lag s) from tape-sync markers -- headless restatement of LagLine
    (util/markers.py:730-790) for projects without azimuth lines: markers are LagSample.to_cfg() tuples
    (a_t, a_f, b_t, b_f, d, corr) (util/markers.py:478-479), a marker sits at t = (a_t + b_t)/2 with lag d.
    The lag is an interpolating spline of order min(smoothing, n-1) through the markers (:737-747), sampled at
    marker rate sr/hop on [0, |duration + lag(duration)|] (:749-758) and band-limited like every BaseLine
    (:601-605; the default band lets everything through).  This is what pytapesynch hands to
    resampling.run(lag_curve=...) (pytapesynch_gui.py:145-155)."""
    from scipy.interpolate import InterpolatedUnivariateSpline
    pts = sorted(((m[0] + m[2]) / 2, m[4]) for m in markers)
    keys = np.array([p[0] for p in pts], dtype=np.float64)
    lags = np.array([p[1] for p in pts], dtype=np.float64)

    def lag_at(times):
        if len(keys) == 0:
            return np.interp(times, (0,), (0,))
        if len(keys) == 1:
            return np.interp(times, keys, lags)
        return InterpolatedUnivariateSpline(keys, lags, k=min(smoothing, len(keys) - 1))(times)

    end = abs(duration + float(lag_at((duration,))[0]))
    marker_sr = sr / hop
    times = np.linspace(0, end, num=int(end * marker_sr))
    lo, hi = sorted(bands)
    return np.stack((times, filters.butter_bandpass_filter(lag_at(times), lo, hi, marker_sr, order=3)), axis=-1)


def correlate_sources(ref_sig, src_sig, sr, lower, upper, ignore_phase=False, window_name=None, speed=1.0):
    """Delay (seconds) and correlation between two signal windows, as the tape-sync tool measures them
    (pytapesynch_gui.py:108-133 behind its widget lookups): both windows band-passed (order-3 zero-phase Butterworth,
    K_sosfiltfilt), then find_delay -- filter outputs stay in HBM and feed the correlation there.  `speed`: the rough
    speed difference the caller resampled `src_sig` by (its match_speed branch uses resampy, not part of this
    package); the delay is corrected for it like the reference does.  The inputs are not modified."""
    from . import correlation, filters
    a_t = filters.bandpass_dev(ref_sig, lower, upper, sr, order=3)
    b_t = filters.bandpass_dev(src_sig, lower, upper, sr, order=3)
    if window_name:                                  # find_delay windows in place: never the caller's own buffers
        a_t = a_t.clone() if a_t is ref_sig else a_t
        b_t = b_t.clone() if b_t is src_sig else b_t
    sample_delay, corr = correlation.find_delay(a_t, b_t, ignore_phase=ignore_phase, window_name=window_name)
    return sample_delay / sr * speed, corr


def tapesync(project, source=None, out_suffix=None, device=None):
    """Run a saved pytapesynch project headless: `project` is the path of a .tapesync JSON (util/widgets.py:
    1224-1233 writes it: fft_size, fft_overlap, markers, source, resampling_mode, sinc_quality, smoothing,
    suffix) or the dict itself; `source` overrides the audio path stored in it.  Writes <source>_res<suffix>.wav
    through resampling.run and returns the lag curve."""
    import json
    from . import io_ops
    cfg = json.load(open(project)) if isinstance(project, (str, bytes)) or hasattr(project, "__fspath__") else dict(project)
    path = source or cfg["source"]
    signal, sr, _ = io_ops.read_file(path)
    hop = cfg["fft_size"] // cfg.get("fft_overlap", 1)
    curve = lag_curve_from_markers(cfg["markers"], len(signal) / sr, sr, hop, cfg.get("smoothing", 3))
    resampling.run((path,), signal_data=((signal, sr),), lag_curve=curve, resampling_mode=cfg.get("resampling_mode", "Sinc"),
                   sinc_quality=cfg.get("sinc_quality", 50), suffix=cfg.get("suffix", "") if out_suffix is None else out_suffix)
    return curve


def heal_project(project, source=None, out_suffix=None, device=None):
    """Run a saved dropout-healer project headless: `project` is a .drop JSON (fft_size, fft_overlap, dropouts,
    surrounding, source, suffix) or the dict itself.  The saved marker tuples carry the two corners in their first
    four fields (a_t, a_f, b_t, b_f, ...); the surrounding factor is the project-wide one.  Writes
    <source>_drops<suffix>.wav like Canvas.resample_files (dropout_healer_gui.py:166) and returns the healed
    (frames, channels) array."""
    import json
    from . import io_ops
    cfg = json.load(open(project)) if isinstance(project, (str, bytes)) or hasattr(project, "__fspath__") else dict(project)
    path = source or cfg["source"]
    signal, sr, channels = io_ops.read_file(path)
    surrounding = cfg.get("surrounding", 0.5)
    marks = [(m[0], m[1], m[2], m[3], surrounding) for m in cfg.get("dropouts", cfg.get("markers", ()))]
    hop = cfg["fft_size"] // cfg.get("fft_overlap", 1)
    healed = heal_dropouts(signal, sr, marks, cfg["fft_size"], hop, device=device)
    io_ops.write_file(path, healed, sr, channels, suffix="_drops" + (cfg.get("suffix", "") if out_suffix is None else out_suffix))
    return healed
