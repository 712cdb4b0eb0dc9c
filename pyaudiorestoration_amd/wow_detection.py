"""Mirror of reference util/wow_detection.py -- same classes, constructor signature and registry.

Track :28-139 (trail sampling and band arithmetic on the host, O(trail)), PeakTracker :294-304,
PeakTrackTracker :307-327, CenterOfGravity :256-291 run their per-frame trace() loops in
K_track (csrc/track.hip) on a device-resident, frame-major float32 magnitude spectrogram.
ZeroCrossingTracker :330-358 band-passes the signal segment with K_sosfiltfilt and finishes the
O(#crossings) post-processing on the host.  CorrelationTracker :396-436 (W4, lowest priority in
SURVEY 8a) is host numpy/scipy exactly like the reference; PartialsTracker needs librosa + a
matplotlib window and is not provided.
"""
import logging
from inspect import isclass

import numpy as np
import scipy.interpolate
import scipy.optimize
import torch
from scipy.signal import get_window

from . import _dev, _lib, filters, fourier
from .correlation import xcorr, parabolic


def nan_helper(y):
    return np.isnan(y), lambda z: z.nonzero()[0]


def interp_nans(y):
    nans, x = nan_helper(y)
    y[nans] = np.interp(x(nans), x(~nans), y[~nans])


def spectrum_to_device(spectrum, dev=None):
    """(bins, frames) numpy/torch spectrogram -> frame-major float32 device tensor [frames][bins].
    Zero-copy when it already is the transposed view K_stft returned."""
    dev = _dev.device_index(dev)
    if isinstance(spectrum, torch.Tensor):
        return spectrum.T.to(device=f"cuda:{dev}", dtype=torch.float32).contiguous()
    return _dev.to_dev(np.asarray(spectrum).T, torch.float32, dev)


class Track:
    tooltip = ""

    def __init__(self, spectrum, signal, trail, fft_size, hop, sr, tolerance_st=1, adaptation_mode="Linear",
                 dB_cutoff=75):
        self.fft_size = fft_size
        self.hop = hop
        self.sr = sr
        self.spectrum = spectrum
        self.signal = signal
        self.fft_freqs = fourier.fft_freqs(fft_size, sr)
        self.frame_0 = 0
        self.num_bins, self.frame_1 = self.spectrum.shape
        self.sample_trail(trail)
        self.NL = 0
        self.NU = 0
        # tolerance in semitones; on log2 scale one semitone is 1/12
        self.tolerance = tolerance_st / 12
        self.min_bins = 4
        self.trace()
        interp_nans(self.freqs)

    def trace(self):
        pass

    def sample_trail(self, trail):
        trail.sort(key=lambda tup: tup[0])
        times_raw = [d[0] for d in trail]
        freqs_raw = [d[1] for d in trail]
        self.ensure_frames(times_raw[0], times_raw[-1])
        self.times = np.linspace(self.frame_0 * self.hop / self.sr, self.frame_1 * self.hop / self.sr,
                                 self.frame_1 - self.frame_0)
        self.freqs = np.interp(self.times, times_raw, freqs_raw)

    def bin_2_freq(self, b):
        return b / self.fft_size * self.sr

    def freq_2_bin(self, f):
        return max(1, min(self.num_bins - 1, int(round(f * self.fft_size / self.sr))))

    def time_2_frame(self, t):
        return int(t * self.sr / self.hop)

    def ensure_frames(self, t0, t1):
        if t0:
            self.frame_0 = max(self.frame_0, self.time_2_frame(t0))
        if t1:
            self.frame_1 = min(self.frame_1, self.time_2_frame(t1))
        if self.frame_0 == self.frame_1:
            logging.warning("No point in tracing just one FFT")

    def set_bin_limits(self, fL, fU):
        fL = max(1.0, fL)
        fU = min(self.sr / 2, fU)
        self.NL = self.freq_2_bin(fL)
        self.NU = self.freq_2_bin(fU)
        while (self.NU - self.NL) < self.min_bins:
            self.NL -= 1
            self.NU += 1

    def freq_plus_tolerance(self, freq, tolerance=None):
        if tolerance is None:
            tolerance = self.tolerance
        logfreq = np.log2(freq)
        return np.power(2, (logfreq - tolerance)), np.power(2, (logfreq + tolerance))

    # -- device plumbing shared by the HIP-backed trackers
    def _run_device(self, kind, mode=0):
        if len(self.freqs) == 0:
            return
        dev = _dev.device_index(self.spectrum.device if isinstance(self.spectrum, torch.Tensor) else None)
        L = _lib.lib()
        mag = spectrum_to_device(self.spectrum, dev)
        n_frames, bins = mag.shape
        f_t = _dev.to_dev(self.freqs, torch.float64, dev)
        if kind == "peak":
            _lib.check(L.par_track_peak_f64(dev, _dev.ptr(mag), n_frames, bins, self.frame_0, len(self.freqs),
                                            _dev.ptr(f_t), self.fft_size, float(self.sr), float(self.tolerance), mode,
                                            _dev.stream_ptr(dev)))
        else:
            _lib.check(L.par_track_cog_f64(dev, _dev.ptr(mag), n_frames, bins, self.frame_0, len(self.freqs),
                                           _dev.ptr(f_t), self.fft_size, float(self.sr), float(self.tolerance),
                                           _dev.stream_ptr(dev)))
        self.freqs[:] = f_t.cpu().numpy()


def fit_sin(tt, yy, assumed_freq=None):
    """Fit a sine to the input sequence (reference util/wow_detection.py:190-228; host, O(len) once per gesture)."""
    tt = np.array(tt)
    yy = np.array(yy)
    ff = np.fft.rfftfreq(len(tt), (tt[1] - tt[0]))
    fft_data = np.fft.rfft(yy)[1:]
    if assumed_freq:
        period = tt[1] - tt[0]
        N = len(yy) + 1
        peak_est = int(round(assumed_freq * N * period))
        win = np.interp(np.arange(0, len(fft_data)), (0, peak_est, len(fft_data)), (0, 1, 0))
        fft_data *= win
    peak_bin = np.argmax(np.abs(fft_data)) + 1
    guess_freq = ff[peak_bin]
    guess_amp = np.std(yy) * 2. ** 0.5
    guess_offset = np.mean(yy)
    guess_phase = np.angle(fft_data[peak_bin])
    guess = np.array([guess_amp, 2. * np.pi * guess_freq, guess_phase, guess_offset])

    def sinfunc(t, A, w, p, c):
        return A * np.sin(w * t + p) + c

    popt, pcov = scipy.optimize.curve_fit(sinfunc, tt, yy, p0=guess)
    A, w, p, c = popt
    f = w / (2. * np.pi)
    return {"amp": A, "omega": w, "phase": p, "offset": c, "freq": f, "period": 1. / f,
            "fitfunc": lambda t: A * np.sin(w * t + p) + c, "maxcov": np.max(pcov), "rawres": (guess, popt, pcov)}


def trace_sine_reg(speed_curve, t0, t1, rpm=None):
    """Regression on an area of the master speed curve (reference util/wow_detection.py:231-253)."""
    times = speed_curve[:, 0]
    speeds = speed_curve[:, 1]
    period = times[1] - times[0]
    ind_start = int(t0 / period)
    ind_stop = int(t1 / period)
    try:
        assumed_freq = float(rpm) / 60
    except Exception:
        assumed_freq = None
    res = fit_sin(times[ind_start:ind_stop], speeds[ind_start:ind_stop], assumed_freq=assumed_freq)
    return res["amp"], res["omega"], res["phase"], 0


class CenterOfGravity(Track):
    name = 'Center of Gravity'

    def trace(self):
        self._run_device("cog")


class PeakTracker(Track):
    name = 'Peak'
    tooltip = "Tracks the mouse input to the loudest peak frequency"

    def trace(self):
        self._run_device("peak", 0)


class PeakTrackTracker(Track):
    name = 'Peak Track'
    tooltip = "Follows the first peak frequency established"

    def trace(self):
        self._run_device("peak", 1)


class ZeroCrossingTracker(Track):
    name = 'Zero-Crossing'
    tooltip = "Track the distance between zero-crossings of the waveform. Good for flutter detection of clean signals"

    def trace(self):
        fL, _ = self.freq_plus_tolerance(np.min(self.freqs))
        _, fU = self.freq_plus_tolerance(np.max(self.freqs))
        s_0 = int(self.times[0] * self.sr)
        s_1 = int(self.times[-1] * self.sr)
        filtered_sig = filters.butter_bandpass_filter(self.signal[s_0:s_1, 0], fL, fU, self.sr, order=3)
        crossings = zero_crossings(filtered_sig)
        deltas = np.diff(crossings).astype(np.float32)
        size = int(self.sr / 100 / np.mean(deltas))
        padded = np.pad(deltas, size, mode='reflect')
        win_sq = get_window("hann", size)
        deltas_conv = np.convolve(padded, win_sq / size * 2, mode="same")[size:-size]
        self.freqs[:] = np.interp(self.times, crossings[:len(deltas_conv)] / self.sr + self.times[0],
                                  self.sr / 2 / deltas_conv)


class PartialsTracker(Track):
    name = 'Partials'

    def trace(self):
        raise NotImplementedError("PartialsTracker needs librosa.piptrack and an interactive matplotlib window "
                                  "(reference util/wow_detection.py:361-387); not part of the HIP hot path")


class FreehandTracker(Track):
    name = 'Freehand Draw'

    def trace(self):
        pass


class CorrelationTracker(Track):
    name = 'Correlation'
    tooltip = "Compare the spectra for each segment and track the offsets between"

    def trace(self):
        spec = self.spectrum.cpu().numpy() if isinstance(self.spectrum, torch.Tensor) else self.spectrum
        fL = min(self.freqs)
        fU = max(self.freqs)
        self.set_bin_limits(fL, fU)
        num_freq_samples = (self.NU - self.NL) * 4
        log_fft_freqs = np.log2(self.fft_freqs[self.NL:self.NU])
        linspace_fft_freqs = np.linspace(log_fft_freqs[0], log_fft_freqs[-1], num_freq_samples)
        resampled = np.ones((num_freq_samples, len(self.freqs) + 1), )
        for i in range(len(self.freqs)):
            interpolator = scipy.interpolate.interp1d(log_fft_freqs, spec[self.NL:self.NU, i], kind='quadratic')
            resampled[:, i] = interpolator(linspace_fft_freqs)
        wind = np.hanning(num_freq_samples)
        changes = np.ones(len(self.freqs))
        for i in range(len(self.freqs)):
            res = xcorr(resampled[:, i] * wind, resampled[:, i + 1] * wind, mode="same")
            i_peak = np.argmax(res)
            i_interp, corr = parabolic(res, i_peak)
            changes[i] = (num_freq_samples // 2) - i_interp
        speed = np.cumsum(changes)
        speed = speed / num_freq_samples * (log_fft_freqs[-1] - log_fft_freqs[0])
        log_mean_freq = np.log2((fL + fU) / 2)
        np.power(2, (log_mean_freq + speed), self.freqs)


class SineRegression(Track):
    name = 'Sine Regression'
    tooltip = "Perform a regression on an area of the master speed curve to yield a sine fit"

    def trace(self):
        pass


def zero_crossings(a):
    positive = a > 0
    return np.where(np.bitwise_xor(positive[1:], positive[:-1]))[0]


wow_detectors = {}
for symbol, value in dict(locals()).items():
    if isclass(value) and value != Track and issubclass(value, Track):
        wow_detectors[value.name] = value
