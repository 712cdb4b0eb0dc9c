"""Wow/flutter pitch trackers behind the reference's tracker API (util/wow_detection.py).

API kept (what pyrespeeder_gui.py:185-189 and the headless pipeline rely on): the `wow_detectors` registry
maps the GUI names to classes constructed as `cls(spectrum, signal, trail, fft_size, hop, sr, tolerance_st,
adaptation_mode)`; construction runs the trace and leaves `times` / `freqs` (Hz per STFT frame of the trail's
span).  `fit_sin` / `trace_sine_reg` keep their return contracts.  Everything else is this package's design:

* trail sampling (frame span, time grid, drawn frequencies) is a pure function, `sample_trail`;
* Peak / Peak Track / Center of Gravity run in K_track (csrc/track.hip) on the device-resident, frame-major
  float32 magnitude spectrogram K_stft wrote -- no spectrogram ever returns to the host;
* Zero-Crossing: band-pass in K_sosfiltfilt, sign-change compaction in K_track (`par_zero_crossings_f64`); only
  the crossing indices come back for the O(#crossings) smoothing;
* Correlation: all frames in one K_track launch sequence (`par_track_corr_f64`): the quadratic-spline resampling as
  a small dense product with the spline's matrix, windowed cross-correlation and parabolic refinement per frame, the
  cumulative sum in order -- the spectrogram stays in HBM;
* Partials (`PartialsTracker`): librosa.piptrack's parabolic peak picking on the device (`par_piptrack_f32`), the
  reference's interactive matplotlib window left out; parity unpinned (librosa is not installed: builder's restatement).

Reference semantics each piece reproduces are cited at the definitions.
"""
import ctypes
import logging

import numpy as np
import scipy.optimize
import torch
from scipy.interpolate import make_interp_spline
from scipy.signal import get_window

from . import _dev, _lib, filters, fourier

MIN_BAND_BINS = 4


# ------------------------------------------------------------------------------------------ helpers

def interp_nans(y):
    """Fill NaNs of a 1-D array in place by linear interpolation over the valid samples (the reference's
    post-processing step, util/wow_detection.py:15-25, 61)."""
    bad = np.isnan(y)
    if bad.any():
        pos = np.arange(len(y))
        y[bad] = np.interp(pos[bad], pos[~bad], y[~bad])


def sample_trail(trail, n_frames, hop, sr):
    """Drawn trail [(t seconds, f Hz), ...] -> (frame_0, frame_1, times, freqs).

    Semantics of Track.sample_trail / ensure_frames (util/wow_detection.py:63-95): the trail is sorted by time
    IN PLACE (callers see it), the frame span is [int(t_first*sr/hop), int(t_last*sr/hop)] clipped to the
    spectrogram -- a bound that is exactly 0 leaves the clip at the spectrogram edge -- and the time grid has one
    point per frame but runs to the END of the span (linspace over frame_1 - frame_0 points)."""
    trail.sort(key=lambda point: point[0])
    t_pts = np.array([p[0] for p in trail], dtype=np.float64)
    f_pts = np.array([p[1] for p in trail], dtype=np.float64)
    frame_0, frame_1 = 0, int(n_frames)
    if t_pts[0]:
        frame_0 = max(frame_0, int(t_pts[0] * sr / hop))
    if t_pts[-1]:
        frame_1 = min(frame_1, int(t_pts[-1] * sr / hop))
    if frame_0 == frame_1:
        logging.warning("No point in tracing just one FFT")
    times = np.linspace(frame_0 * hop / sr, frame_1 * hop / sr, frame_1 - frame_0)
    return frame_0, frame_1, times, np.interp(times, t_pts, f_pts)


def octave_band(freq, half_width_octaves):
    """(f * 2^-w, f * 2^+w): the tolerance band around a frequency (util/wow_detection.py:109-117)."""
    centre = np.log2(freq)
    return np.power(2, centre - half_width_octaves), np.power(2, centre + half_width_octaves)


def band_to_bins(f_lo, f_hi, fft_size, sr, n_bins, min_bins=MIN_BAND_BINS):
    """Band edges in Hz -> (NL, NU) bin slice, at least `min_bins` wide (util/wow_detection.py:81-82, 97-107:
    edges clamped to [1 Hz, Nyquist], bins to [1, n_bins-1], Python round-half-even, symmetric widening)."""
    def to_bin(f):
        return int(min(n_bins - 1, max(1, int(np.rint(f * fft_size / sr)))))
    lo, hi = to_bin(max(1.0, f_lo)), to_bin(min(sr / 2, f_hi))
    missing = min_bins - (hi - lo)
    if missing > 0:
        grow = (missing + 1) // 2
        lo, hi = lo - grow, hi + grow
    return lo, hi


def spectrum_to_device(spectrum, dev=None):
    """(bins, frames) numpy/torch spectrogram -> frame-major float32 device tensor [frames][bins] whose rows may be more than
    `bins` floats apart (stride(0) = the pitch K_stft wrote with, stride(1) = 1).  Zero-copy when it already is the
    transposed view K_stft returned."""
    dev = _dev.device_index(dev)
    if isinstance(spectrum, torch.Tensor):
        fm = spectrum.T
        if fm.device.type == "cuda" and fm.dtype == torch.float32 and fm.ndim == 2 and fm.stride(1) == 1 and fm.stride(0) >= fm.shape[1]:
            return fm
        return fm.to(device=f"cuda:{dev}", dtype=torch.float32).contiguous()
    return _dev.to_dev(np.asarray(spectrum).T, torch.float32, dev)


def zero_crossings_dev(x_t, dev=None):
    """Indices i with (x[i+1] > 0) != (x[i] > 0) of a float64 device tensor, ascending, as a device int64
    tensor (zero_crossings, util/wow_detection.py:448-450)."""
    dev = _dev.device_index(dev if dev is not None else x_t.device)
    L = _lib.lib()
    n = x_t.numel()
    work = _dev.empty(int(L.par_zero_crossings_work_len(n)), torch.int64, dev)
    count = ctypes.c_int64(0)
    _lib.check(L.par_zero_crossings_f64(dev, _dev.ptr(x_t), n, _dev.ptr(work), None, 0, ctypes.byref(count),
                                        _dev.stream_ptr(dev)))
    idx = _dev.empty(count.value, torch.int64, dev)
    if count.value:
        _lib.check(L.par_zero_crossings_f64(dev, _dev.ptr(x_t), n, _dev.ptr(work), _dev.ptr(idx), idx.numel(),
                                            ctypes.byref(count), _dev.stream_ptr(dev)))
    return idx


def zero_crossings(a):
    """Host-array convenience form of zero_crossings_dev (same name as the reference's helper)."""
    dev = _dev.device_index(None)
    return zero_crossings_dev(_dev.to_dev(a, torch.float64, dev), dev).cpu().numpy()


# ------------------------------------------------------------------------------------------ trackers

class Track:
    """Base of all trackers: samples the trail, runs `trace()`, patches NaNs.  Attributes the callers read:
    times, freqs (and frame_0 / frame_1, the spectrogram frames they refer to)."""
    name = None
    tooltip = ""

    def __init__(self, spectrum, signal, trail, fft_size, hop, sr, tolerance_st=1, adaptation_mode="Linear",
                 dB_cutoff=75, refine=None):
        # refine (optional, beyond the reference's signature): {"x": 1-D float32 device tensor (the channel the spectrogram
        # was made from, possibly a strided view), "n_fft", "zeropad", "window": float32 device tensor[n_fft]} -- Peak and
        # Peak Track then re-evaluate their band magnitudes from the signal in float64 (par_track_peak_refined_f64), which
        # is what the reference's numpy backend hands them; without it they read the float32 spectrogram
        self.refine = refine
        self.spectrum, self.signal = spectrum, signal
        self.fft_size, self.hop, self.sr = fft_size, hop, sr
        self.num_bins, n_frames = spectrum.shape
        self.fft_freqs = fourier.fft_freqs(fft_size, sr)
        self.frame_0, self.frame_1, self.times, self.freqs = sample_trail(trail, n_frames, hop, sr)
        self.tolerance = tolerance_st / 12                 # semitones -> octaves
        self.NL = self.NU = 0
        self.trace()
        interp_nans(self.freqs)

    def trace(self):
        """Overwrite self.freqs in place; the base class keeps the drawn trail."""

    def _device_spectrum(self):
        dev = _dev.device_index(self.spectrum.device if isinstance(self.spectrum, torch.Tensor) else None)
        return dev, spectrum_to_device(self.spectrum, dev)

    def _trace_on_device(self, kernel, *extra, needs_first=False):
        """Per-frame band search in K_track: freqs go up as the drawn trail and come back traced.  needs_first:
        the tracker seeds its band from freqs[0], so an empty span is the reference's IndexError."""
        if len(self.freqs) == 0:
            if needs_first:
                raise IndexError("index 0 is out of bounds for axis 0 with size 0")
            return
        dev, mag = self._device_spectrum()
        f_t = _dev.to_dev(self.freqs, torch.float64, dev)
        status = _dev.empty(1, torch.int32, dev)              # "empty band" word: ParEmptyBand (a ValueError) if set
        _lib.check(kernel(dev, _dev.ptr(mag), mag.shape[0], mag.shape[1], mag.stride(0), self.frame_0, len(self.freqs), _dev.ptr(f_t),
                          self.fft_size, float(self.sr), float(self.tolerance), *extra, _dev.ptr(status),
                          _dev.stream_ptr(dev)))
        self.freqs[:] = f_t.cpu().numpy()

    def _trace_refined(self, mode, needs_first=False):
        """Band magnitudes from the signal in float64 when the caller supplied `refine`; False -> use the spectrogram."""
        rf = self.refine
        if not rf:
            return False
        if len(self.freqs) == 0:
            if needs_first:
                raise IndexError("index 0 is out of bounds for axis 0 with size 0")
            return True
        x_t, win_t = rf["x"], rf["window"]
        dev = _dev.device_index(x_t.device)
        n_fft, zp = int(rf["n_fft"]), int(rf.get("zeropad", 1))
        if n_fft * zp != self.fft_size or x_t.ndim != 1:
            return False
        # the float64 band re-evaluation is a direct DFT: bins x n_fft per frame.  A wide tolerance on a long, zero-padded
        # transform would turn milliseconds into seconds: such traces read the spectrogram instead
        f_hi = float(np.max(self.freqs)) if len(self.freqs) else 0.0
        band_bins = max(4.0, f_hi * (2.0 ** self.tolerance - 2.0 ** -self.tolerance) * self.fft_size / float(self.sr)) + 2.0
        if band_bins * n_fft * len(self.freqs) > 2.0e10:
            return False
        stride = x_t.stride(0)
        f_t = _dev.to_dev(self.freqs, torch.float64, dev)
        status = _dev.empty(1, torch.int32, dev)
        try:
            _lib.check(_lib.lib().par_track_peak_refined_f64(
                dev, _dev.ptr(x_t), x_t.shape[0], stride, n_fft, self.hop, zp, _dev.ptr(win_t), self.num_bins,
                self.spectrum.shape[1], self.frame_0, len(self.freqs), _dev.ptr(f_t), float(self.sr), float(self.tolerance),
                mode, _dev.ptr(status), _dev.stream_ptr(dev)))
        except _lib.ParUnsupported:
            return False
        self.freqs[:] = f_t.cpu().numpy()
        return True


class CenterOfGravity(Track):
    name = 'Center of Gravity'

    def trace(self):          # util/wow_detection.py:256-291 -> k_track_cog
        self._trace_on_device(_lib.lib().par_track_cog_f64, needs_first=True)


class PeakTracker(Track):
    name = 'Peak'
    tooltip = "Tracks the mouse input to the loudest peak frequency"

    def trace(self):          # util/wow_detection.py:294-304 -> k_track_peak (or k_track_peak_refined)
        if not self._trace_refined(0):
            self._trace_on_device(_lib.lib().par_track_peak_f64, 0)

class PeakTrackTracker(Track):
    name = 'Peak Track'
    tooltip = "Follows the first peak frequency established"

    def trace(self):          # util/wow_detection.py:307-327 -> k_track_peak_fixed
        if not self._trace_refined(1, needs_first=True):
            self._trace_on_device(_lib.lib().par_track_peak_f64, 1, needs_first=True)


def crossing_periods_to_freqs(crossings, sr, t_first, times):
    """Sign-change indices -> frequency per requested time (util/wow_detection.py:342-358): the spacing of
    consecutive crossings (half periods, float32 like the reference) is smoothed with a Hann kernel of about
    10 ms of crossings on a reflect-padded sequence, turned into Hz and interpolated onto `times`."""
    half_periods = np.diff(crossings).astype(np.float32)
    span = int(sr / 100 / np.mean(half_periods))
    kernel = get_window("hann", span) / span * 2
    smooth = np.convolve(np.pad(half_periods, span, mode='reflect'), kernel, mode="same")[span:-span]
    return np.interp(times, crossings[:len(smooth)] / sr + t_first, sr / 2 / smooth)


class ZeroCrossingTracker(Track):
    name = 'Zero-Crossing'
    tooltip = "Track the distance between zero-crossings of the waveform. Good for flutter detection of clean signals"

    def trace(self):          # util/wow_detection.py:330-358
        f_lo = octave_band(np.min(self.freqs), self.tolerance)[0]
        f_hi = octave_band(np.max(self.freqs), self.tolerance)[1]
        first, last = int(self.times[0] * self.sr), int(self.times[-1] * self.sr)
        band = filters.bandpass_dev(self.signal[first:last, 0], f_lo, f_hi, self.sr, order=3)
        crossings = zero_crossings_dev(band).cpu().numpy()
        self.freqs[:] = crossing_periods_to_freqs(crossings, self.sr, self.times[0], self.times)


def piptrack_dev(mag_t, fft_size, sr, fmin, fmax, threshold=0.1, scale=None, offset=1e-7, dev=None):
    """librosa.piptrack on a device magnitude spectrogram as fourier.get_mag returns it ((bins, frames) view of the
    frame-major buffer): (pitches, magnitudes), both (bins, frames) float32 device tensors.  `scale` / `offset` undo
    get_mag's 1/sqrt(n_fft) and + 1e-7 (defaults), so the magnitudes are those of librosa's own |stft|."""
    dev = _dev.device_index(dev if dev is not None else mag_t.device)
    fm = mag_t.T                                            # frame-major [frames][bins], rows stride(0) floats apart
    if not (fm.stride(1) == 1 and fm.stride(0) >= fm.shape[1]):
        fm = fm.contiguous()
    frames, bins = fm.shape
    pitches = _dev.empty((frames, bins), torch.float32, dev)
    mags = _dev.empty((frames, bins), torch.float32, dev)
    _lib.check(_lib.lib().par_piptrack_f32(dev, _dev.ptr(fm), frames, bins, fm.stride(0), float(np.sqrt(fft_size) if scale is None else scale),
                                           float(offset), int(fft_size), float(sr), float(fmin), float(fmax), float(threshold),
                                           _dev.ptr(pitches), _dev.ptr(mags), _dev.stream_ptr(dev)))
    return pitches.T, mags.T


class PartialsTracker(Track):
    """util/wow_detection.py:361-387: the reference computes librosa.piptrack(y=signal[:, 0], fmin=min(trail),
    fmax=max(trail), threshold=0.15) and SHOWS the pitch map in a matplotlib window -- it never writes self.freqs, so the
    traced line is the drawn trail.  Here the map is computed on the device (K_stft with librosa's Hann window, then
    par_piptrack_f32) and kept as .pitches / .magnitudes ((bins, frames) device tensors); no window is opened.  The
    signal is zero-padded like librosa >= 0.10's centred STFT (where hop divides fft_size / 2)."""
    name = 'Partials'

    def trace(self):
        if len(self.freqs) == 0:
            raise ValueError("zero-size array to reduction operation minimum which has no identity")   # np.min(self.freqs)
        fl, fu = float(np.min(self.freqs)), float(np.max(self.freqs))
        dev = _dev.device_index(self.spectrum.device if isinstance(self.spectrum, torch.Tensor) else None)
        sig = np.ascontiguousarray(np.asarray(self.signal)[:, 0], dtype=np.float32)
        half = self.fft_size // 2
        if half % self.hop == 0:
            # librosa >= 0.10 centres its frames on a ZERO-padded signal (pad_mode="constant"); K_stft reflects.  With the
            # zeros put there explicitly, frame f of the signal is frame f + half/hop of the padded one, whose window never
            # reaches the padded signal's own (reflected) ends
            padded = _dev.empty(len(sig) + 2 * half, torch.float32, dev)
            padded.zero_()
            padded[half:half + len(sig)] = _dev.to_dev(sig, torch.float32, dev)
            full = fourier.get_mag(padded, self.fft_size, self.hop, "hann", 1)
            skip = half // self.hop
            mag = full[:, skip:skip + len(sig) // self.hop + 1]
        else:                                           # frames of the padded signal do not line up: reflected edges (two frames differ)
            mag = fourier.get_mag(_dev.to_dev(sig, torch.float32, dev), self.fft_size, self.hop, "hann", 1)
        self.pitches, self.magnitudes = piptrack_dev(mag, self.fft_size, self.sr, fl, fu, threshold=0.15, dev=dev)
        logging.info("Partials: pitch / magnitude maps are in .pitches / .magnitudes; the interactive plot is not shown")


class FreehandTracker(Track):
    name = 'Freehand Draw'     # returns the interpolated trail itself


def correlation_spline_matrix(log_freqs, n_grid=None):
    """The quadratic interpolating spline of `interp1d(log_freqs, y, kind='quadratic')` evaluated on the reference's
    4x oversampled uniform grid, as a matrix: grid values = M @ band values (the spline is linear in y and its
    abscissae are the same for every frame, so scipy's own construction is applied once, to the identity).
    n_grid: grid points when they are not 4 per band value (a band clipped at the last bin keeps the grid of the
    unclipped one, util/wow_detection.py:404-408)."""
    nb = len(log_freqs)
    grid = np.linspace(log_freqs[0], log_freqs[-1], 4 * nb if n_grid is None else n_grid)
    return np.ascontiguousarray(make_interp_spline(log_freqs, np.eye(nb), k=2, axis=0, check_finite=False)(grid))


class CorrelationTracker(Track):
    name = 'Correlation'
    tooltip = "Compare the spectra for each segment and track the offsets between"

    def trace(self):          # util/wow_detection.py:396-436 (reads frames 0.. whatever frame_0 is: kept) -> k_corr_*
        f_lo, f_hi = float(np.min(self.freqs)), float(np.max(self.freqs))
        self.NL, self.NU = band_to_bins(f_lo, f_hi, self.fft_size, self.sr, self.num_bins)
        count = len(self.freqs)
        if count == 0:
            return
        if self.NL < 0 or count > self.spectrum.shape[1]:
            raise IndexError(f"index {count - 1} is out of bounds for axis 1 with size {self.spectrum.shape[1]}"
                             if self.NL >= 0 else "correlation band reaches below bin 0")
        log_f = np.log2(self.fft_freqs[self.NL:self.NU])
        n = 4 * (self.NU - self.NL)
        dev, mag = self._device_spectrum()
        L = _lib.lib()
        M_t = _dev.to_dev(correlation_spline_matrix(log_f, n), torch.float64, dev)
        w_t = _dev.to_dev(np.hanning(n), torch.float64, dev)
        work = _dev.empty(int(L.par_track_corr_work_len(count, n)), torch.float64, dev)
        f_t = _dev.empty(count, torch.float64, dev)
        status = _dev.empty(1, torch.int32, dev)
        _lib.check(L.par_track_corr_f64(dev, _dev.ptr(mag), mag.shape[0], mag.shape[1], mag.stride(0), self.NL, self.NU, count, _dev.ptr(M_t),
                                        _dev.ptr(w_t), n, float(log_f[-1] - log_f[0]), float(np.log2((f_lo + f_hi) / 2)),
                                        _dev.ptr(work), _dev.ptr(f_t), _dev.ptr(status), _dev.stream_ptr(dev)))
        self.freqs[:] = f_t.cpu().numpy()


class SineRegression(Track):
    name = 'Sine Regression'
    tooltip = "Perform a regression on an area of the master speed curve to yield a sine fit"


# ---------------------------------------------------------------------------------- sine regression

def fit_sin(tt, yy, assumed_freq=None):
    """Least-squares fit of A*sin(w*t + p) + c to uniformly sampled data; returns the reference's result dict
    (util/wow_detection.py:190-228: "amp", "omega", "phase", "offset", "freq", "period", "fitfunc", "maxcov",
    "rawres").  The start point comes from the strongest non-DC FFT bin, optionally tapered towards an assumed
    frequency; as in the reference the start phase is read one bin above that peak (its DC-less indexing)."""
    t = np.array(tt)
    y = np.array(yy)
    step = t[1] - t[0]
    bins = np.fft.rfft(y)[1:]                                  # DC dropped: bins[j] is frequency bin j + 1
    if assumed_freq:
        centre = int(round(assumed_freq * (len(y) + 1) * step))
        bins = bins * np.interp(np.arange(len(bins)), (0, centre, len(bins)), (0, 1, 0))
    k = int(np.argmax(np.abs(bins))) + 1
    start = np.array([np.std(y) * 2. ** 0.5, 2. * np.pi * np.fft.rfftfreq(len(t), step)[k], np.angle(bins[k]), np.mean(y)])

    def model(x, amp, omega, phase, offset):
        return amp * np.sin(omega * x + phase) + offset

    best, cov = scipy.optimize.curve_fit(model, t, y, p0=start)
    amp, omega, phase, offset = best
    freq = omega / (2. * np.pi)
    return {"amp": amp, "omega": omega, "phase": phase, "offset": offset, "freq": freq, "period": 1. / freq,
            "fitfunc": lambda x: model(x, *best), "maxcov": np.max(cov), "rawres": (start, best, cov)}


def trace_sine_reg(speed_curve, t0, t1, rpm=None):
    """Sine fit over [t0, t1] of a master speed curve (N, 2) -> (amp, omega, phase, 0)
    (util/wow_detection.py:231-253); rpm, if it parses as a number, sets the assumed wow frequency rpm/60."""
    step = speed_curve[1, 0] - speed_curve[0, 0]
    part = speed_curve[int(t0 / step):int(t1 / step)]
    try:
        wow_hz = float(rpm) / 60
    except (TypeError, ValueError):
        wow_hz = None
    fit = fit_sin(part[:, 0], part[:, 1], assumed_freq=wow_hz)
    return fit["amp"], fit["omega"], fit["phase"], 0


wow_detectors = {cls.name: cls for cls in Track.__subclasses__()}
