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

import numpy as np
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


def respeed(signal, sr, trail, fft_size=1024, hop=256, zeropad=1, mode="Peak", tolerance_st=0.5, bands=(0, 20),
            sinc_quality=32, resampling_mode="Sinc", device=None):
    """signal: float32 (n,) or (n, ch).  Returns dict with every intermediate the reference's GUI
    would hold: spectrum (device), track times/freqs, speed curve, positions (device), output."""
    dev = _dev.device_index(device)
    sig2d = signal[:, None] if signal.ndim == 1 else signal
    n = sig2d.shape[0]
    sig_t = _dev.to_dev(sig2d, torch.float32, dev)                       # (n, ch) resident in HBM
    ch = sig2d.shape[1]
    spec = fourier.get_mag(sig_t.reshape(-1)[0::ch] if ch > 1 else sig_t.reshape(-1), fft_size, hop,
                           "blackmanharris", zeropad)                      # device tensor (bins, frames)
    track = wow_detection.wow_detectors[mode](spec, sig2d, list(trail), fft_size * zeropad, hop, sr, tolerance_st,
                                              "Linear")
    curve = master_speed_curve([(track.times, trace_to_speed(track.freqs))], n / sr, sr, hop, bands)
    st_t = _dev.to_dev(curve[:, 0] * sr, torch.float64, dev)
    sp_t = _dev.to_dev(np.ascontiguousarray(curve[:, 1]), torch.float64, dev)
    pos_t = resampling.speed_to_pos_dev(st_t, sp_t, n, dev)      # kept: the GUI shows / reuses sample_at
    plan = resampling.speed_plan_dev(st_t, sp_t, n, dev, fused=True) if resampling_mode == "Sinc" else None
    out_t = _dev.empty((pos_t.numel(), ch), torch.float32, dev)
    for c in range(ch):
        if resampling_mode == "Sinc" and plan.fused_ok:
            resampling.varispeed_fused_dev(plan, sig_t.reshape(-1)[c:], sinc_quality, out_t.reshape(-1)[c:], sig_stride=ch,
                                           len_in=n, out_stride=ch)
        elif resampling_mode == "Sinc":
            resampling.sinc_resample_dev(pos_t, sig_t.reshape(-1)[c:], sinc_quality, out_t.reshape(-1)[c:], sig_stride=ch,
                                         len_in=n, out_stride=ch, dev=dev)
        else:
            resampling.linear_resample_dev(pos_t, sig_t.reshape(-1)[c:], out_t.reshape(-1)[c:], sig_stride=ch, len_in=n,
                                           out_stride=ch, dev=dev)
    return {"spectrum": spec, "times": track.times, "freqs": track.freqs, "speed_curve": curve, "positions": pos_t,
            "output": out_t}


# ---------------------------------------------------------------------- config 4: dropout healer
def to_dB(a):
    """util/units.py:24-25."""
    return 20 * np.log10(a)


def heal_dropouts(signal, sr, markers, fft_size=512, hop=32, channels=None, device=None):
    """Spectral inpainting of marked dropouts -- headless restatement of
    dropout_healer_gui.Canvas.resample_files (dropout_healer_gui.py:111-166).

    signal: float32 (n, ch).  markers: iterable of (a0, a1, b0, b1, surrounding) = the reference's
    DropoutSample.to_cfg() (util/markers.py:368-388, 424-426): corner (t, f) pairs and the
    surrounding factor.  STFT, gain application and ISTFT run on the device; the per-marker target
    (mean dB of the frames before/after, bilinear fill, clip against earlier markers) is O(box) host
    math on slices copied back from HBM, like the GUI does it on its cached spectrogram."""
    import ctypes
    from scipy.interpolate import RegularGridInterpolator
    from . import _lib
    dev = _dev.device_index(device)
    L = _lib.lib()
    sig2d = signal[:, None] if signal.ndim == 1 else signal
    n, ch = sig2d.shape
    if channels is None:
        channels = range(ch)
    out = np.empty(sig2d.shape, dtype=sig2d.dtype)
    y_pad = fourier.fix_length(sig2d, n + fft_size // 2, axis=0)
    pad_t = _dev.to_dev(y_pad, torch.float32, dev)                      # (n + fft/2, ch) in HBM

    def t2f(t):
        return int(t * sr / hop)

    def f2b(f):
        return max(1, min(fft_size // 2, int(round(f * fft_size / sr))))

    for c in channels:
        S = fourier.stft(pad_t.reshape(-1)[c::ch] if ch > 1 else pad_t.reshape(-1), n_fft=fft_size, step=hop)  # (bins, frames) device
        fm = S.T                                                           # frame-major [frames][bins] view, contiguous
        gain = torch.zeros(fm.shape, dtype=torch.float32, device=fm.device)
        for (a0, a1, b0, b1, surrounding) in markers:
            width, t = abs(a0 - b0), (a0 + b0) / 2
            f, height = (a1 + b1) / 2, abs(a1 - b1)
            frame_b, frame_a = t2f(t - width / 2), t2f(t + width / 2)
            fs = max(1, t2f(width * surrounding))
            bin_l, bin_u = f2b(f - height / 2), f2b(f + height / 2)
            box = fm[frame_b - fs:frame_a + fs, bin_l:bin_u].cpu().numpy()      # small D2H
            db = to_dB(np.abs(box.astype(np.complex128)) + .0000001).T           # (bins, frames) like the reference
            mag_before = np.mean(db[:, 0:fs], axis=1)
            mag_after = np.mean(db[:, fs + (frame_a - frame_b):fs + (frame_a - frame_b) + fs], axis=1)
            fp_frames = np.linspace(frame_b, frame_a, num=frame_a - frame_b)
            fp_bins = np.linspace(bin_l, bin_u, num=bin_u - bin_l)
            interp = RegularGridInterpolator(((frame_b, frame_a), fp_bins), (mag_before, mag_after))
            mp_bins, mp_frames = np.meshgrid(fp_bins, fp_frames)
            fp_db = np.swapaxes(interp((mp_frames, mp_bins)), 0, 1)
            gain_db = fp_db - db[:, fs:fs + (frame_a - frame_b)]
            prev = gain[frame_b:frame_a, bin_l:bin_u].cpu().numpy().T.astype(np.float64)
            np.clip(gain_db, prev, 255, out=gain_db)
            gain[frame_b:frame_a, bin_l:bin_u] = torch.from_numpy(np.ascontiguousarray(gain_db.T, dtype=np.float32)).to(gain.device)
        healed = fm.contiguous()
        _lib.check(L.par_spec_apply_gain_db_c64(dev, _dev.ptr(healed), _dev.ptr(gain), healed.numel(), _dev.stream_ptr(dev)))
        y = fourier.istft(healed.T, length=n, hop_length=hop)
        out[:, c] = y.cpu().numpy()
    return out
