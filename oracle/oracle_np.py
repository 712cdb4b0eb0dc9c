"""numpy restatement of the reference's util/ hot path -- TEST INFRASTRUCTURE ONLY.

This module is the *checker* for the HIP product path.  Nothing under
``pyaudiorestoration_amd/`` may import it; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.

Parity status: PINNED.  Every function below is checked in
``tests/test_oracle_golden.py`` against golden vectors captured by importing the
real reference (``oracle/gen_golden.py``, run in the build container where
``/root/reference`` exists; numpy 2.2.6 + scipy 1.15.3 are the de-facto pin of the
reference's un-versioned third-party arithmetic).
ONE EXCEPTION, parity UNPINNED: ``piptrack`` restates librosa's published algorithm
(the reference's PartialsTracker calls librosa, which is neither in the reference
checkout nor installed here); it is checked by hand-computed cases only.

Each function cites the reference lines it restates (paths relative to the
reference checkout).  Third-party arithmetic the reference itself delegates to
(``numpy.fft``, ``scipy.signal.get_window/butter/sosfiltfilt/correlate``,
``scipy.interpolate.interp1d``) is called directly here as well: it is the same
dependency, not reference code.
"""
import numpy as np
import scipy.signal
import scipy.interpolate


# --------------------------------------------------------------------------- STFT

def frame_count(n, n_fft, hop):
    """S1  util/fourier.py:78-82 -- frames of a reflect-padded signal (integer exact)."""
    return (n + 2 * (n_fft // 2) - n_fft) // hop + 1


def stft(x, n_fft=1024, step=512, window_name="blackmanharris", zeropad=1):
    """S0-S3  util/fourier.py:37-75 with the numpy backend :136-157.

    reflect-pad n_fft//2, frame i = window * xpad[i*step : i*step+n_fft], zero
    extension at the END up to n_fft*zeropad, rfft in float32 (pocketfft), then a
    float64 division by sqrt(n_fft) (so: complex128 container, c64-precision values).
    """
    n_fft = int(n_fft)
    step = max(n_fft // 2, 1) if step is None else int(step)
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("x must be 1D")
    win = scipy.signal.get_window(window_name, n_fft).astype(np.float32)
    xp = np.pad(x, n_fft // 2, mode="reflect")
    n_frames = (len(xp) - n_fft) // step + 1
    idx = np.arange(n_frames)[:, None] * step + np.arange(n_fft)[None, :]
    frames = (win[None, :] * xp[idx]).astype(np.float32)
    spec = np.fft.rfft(frames, n=n_fft * zeropad, axis=1).T
    return spec / np.sqrt(n_fft)


def to_mag(spec):
    """S4  util/fourier.py:23-24."""
    return np.abs(spec) + 0.0000001


def get_mag(*a, **k):
    """S4  util/fourier.py:27-29."""
    return to_mag(stft(*a, **k))


def fft_freqs(n_fft, fs):
    """S5  util/fourier.py:690-700."""
    return np.arange(0, n_fft // 2 + 1) / float(n_fft) * float(fs)


def fix_length(data, size):
    """S5  util/fourier.py:440-478 (1-D / leading-axis use only)."""
    data = np.asarray(data)
    n = data.shape[0]
    if n > size:
        return data[:size]
    if n < size:
        pad = [(0, size - n)] + [(0, 0)] * (data.ndim - 1)
        return np.pad(data, pad, mode="constant")
    return data


def window_sumsquare(window_name, n_frames, hop_length, n_fft, dtype=np.float64):
    """util/fourier.py:481-546 (win_length == n_fft, norm=None)."""
    n = n_fft + hop_length * (n_frames - 1)
    env = np.zeros(n, dtype=dtype)
    wsq = scipy.signal.get_window(window_name, n_fft) ** 2
    for f in range(n_frames):
        s = f * hop_length
        env[s:min(n, s + n_fft)] += wsq[:max(0, min(n_fft, n - s))]
    return env


def istft(stft_matrix, hop_length=None, window_name="blackmanharris", length=None):
    """S6  util/fourier.py:314-437 (center=True, win_length=n_fft).

    Unlike the reference this does NOT mutate its argument (quirk 9); the
    denormalisation by sqrt(n_fft) is applied to a copy.
    """
    S = np.array(stft_matrix) * 1  # copy
    n_fft = 2 * (S.shape[0] - 1)
    S = S * np.sqrt(n_fft)
    if hop_length is None:
        hop_length = n_fft // 4
    win = scipy.signal.get_window(window_name, n_fft, fftbins=True)
    if length:
        n_frames = min(S.shape[1], int(np.ceil((length + n_fft) / hop_length)))
    else:
        n_frames = S.shape[1]
    real_dtype = np.float32 if S.dtype == np.complex64 else np.float64
    y = np.zeros(n_fft + hop_length * (n_frames - 1), dtype=real_dtype)
    frames = win[:, None] * np.fft.irfft(S[:, :n_frames], axis=0)
    for f in range(n_frames):
        y[f * hop_length:f * hop_length + n_fft] += frames[:, f]
    env = window_sumsquare(window_name, n_frames, hop_length, n_fft, dtype=real_dtype)
    nz = env > np.finfo(env.dtype).tiny
    y[nz] /= env[nz]
    if length is None:
        return y[n_fft // 2:-(n_fft // 2)]
    return fix_length(y[n_fft // 2:], length)


# --------------------------------------------------------------------- resampling

def speed_to_pos(sampletimes, speeds, num_input_samples):
    """R1  util/resampling.py:93-137.

    Returns the *written prefix* (quirk 2): when no segment straddles
    ``num_input_samples`` the reference returns a buffer whose tail is
    uninitialised memory; the defined result is the first sum(n_i) entries.
    The second return value says whether the trim branch fired.
    """
    sampletimes = np.asarray(sampletimes, dtype=np.float64)
    speeds = np.asarray(speeds, dtype=np.float64)
    periods = np.diff(sampletimes)
    err = 0.0
    offset = sampletimes[0]
    cap = int(np.mean(speeds) * (sampletimes[-1] - sampletimes[0]) * 1.01)
    out = np.empty(cap)
    w = 0
    for i in range(len(speeds) - 1):
        want = periods[i] * np.mean(speeds[i:i + 2]) + err
        n = int(round(want))
        err = want - n
        ramp = np.arange(n) / (n - 1) * (speeds[i + 1] - speeds[i]) + speeds[i]
        seg = np.cumsum(1 / ramp) + offset
        offset = seg[-1]
        out[w:w + n] = seg
        if out[w] <= num_input_samples <= out[w + n - 1]:
            end = w + int(np.argmin(np.abs(seg - num_input_samples)))
            return out[:end].copy(), True
        w += n
    return out[:w].copy(), False


def sinc_resample(sample_at, signal, NT, chunk=1 << 15):
    """R2+R3  util/resampling.py:21-27, 51-90 (sinc_wrapper -> sinc_core), vectorised.

    Canonical delta definition (quirk 3): period_to[i] = max(1e-12, p[i+1]-p[i])
    for every i but the last, which reuses the previous one -- exactly what the
    single-threaded ``sinc_wrapper`` does.  Tap k of output i multiplies
    ``signal[lower+k]`` (quirk 1: bug-compatible leading edge).  All math float64,
    result cast to float32.
    """
    p = np.asarray(sample_at, dtype=np.float64)
    sig = np.asarray(signal)
    len_in = len(sig)
    n_out = len(p)
    if n_out < 2:
        raise ValueError("reference raises UnboundLocalError for len_out < 2")
    N = np.arange(-NT, NT + 1, dtype=np.float32)[:2 * NT].astype(np.float64)
    win = np.hanning(2 * NT + 1).astype(np.float32)[:2 * NT].astype(np.float64)
    dp = np.empty(n_out)
    dp[:-1] = np.maximum(1e-12, p[1:] - p[:-1])
    dp[-1] = dp[-2]
    fc = np.minimum(1.0 / dp, 1.0)
    out = np.empty(n_out, dtype=np.float32)
    k = np.arange(2 * NT)
    for a in range(0, n_out, chunk):
        b = min(n_out, a + chunk)
        pc = p[a:b]
        ind = np.rint(pc).astype(np.int64)
        lower = np.maximum(0, ind - NT)
        upper = np.minimum(ind + NT, len_in)
        shift = pc - ind
        idx = lower[:, None] + k[None, :]
        live = idx < upper[:, None]
        taps = np.where(live, sig[np.minimum(idx, len_in - 1)].astype(np.float64), 0.0)
        w = np.sinc((N[None, :] - shift[:, None]) * fc[a:b, None]) * fc[a:b, None]
        out[a:b] = np.sum(taps * w * win[None, :], axis=1)
    return out


def linear_resample(sample_at, signal):
    """R4 'Linear' mode  util/resampling.py:229."""
    return np.interp(sample_at, np.arange(len(signal)), signal, left=0.0, right=0.0).astype(np.float32)


def lag_to_positions(lag_curve, sr, len_signal):
    """R4 lag-curve branch  util/resampling.py:189-206."""
    sampletimes = lag_curve[:, 0] * sr
    lags = lag_curve[:, 1] * sr
    num_out = len_signal + abs(lags[-1])
    pos = np.interp(np.arange(num_out), sampletimes, sampletimes - lags)
    hit = np.nonzero(pos >= len_signal)[0]
    if len(hit):
        pos = pos[:hit[0]]
    return np.clip(pos, 0, None)


# ------------------------------------------------------------ correlation / filters

def parabolic(f, x):
    """X1  util/correlation.py:42-46."""
    xv = 1 / 2. * (f[x - 1] - f[x + 1]) / (f[x - 1] - 2 * f[x] + f[x + 1]) + x
    yv = f[x] - 1 / 4. * (f[x - 1] - f[x + 1]) * (xv - x)
    return xv, yv


def xcorr(a, b, mode="full"):
    """X2  util/correlation.py:6-13."""
    a = a / np.linalg.norm(a)
    b = b / np.linalg.norm(b)
    return scipy.signal.correlate(a, b, mode=mode, method="auto")


def find_delay(a, b, ignore_phase=False, window_name=None):
    """X2  util/correlation.py:16-39 (does not mutate a/b)."""
    a = np.array(a, dtype=np.float64)
    b = np.array(b, dtype=np.float64)
    if window_name:
        a = a * scipy.signal.get_window(window_name, len(a))
        b = b * scipy.signal.get_window(window_name, len(b))
    res = xcorr(a, b, mode="same")
    peak = np.argmax(np.abs(res)) if ignore_phase else np.argmax(res)
    i_peak, corr = parabolic(res, peak)
    return i_peak - len(res) // 2, corr


def butter_bandpass_filter(data, lowcut, highcut, fs, order=5):
    """F1  util/filters.py:7-24."""
    nyq = 0.5 * fs
    low, high = lowcut / nyq, highcut / nyq
    lo_ok, hi_ok = 0 < low < 1, 0 < high < 1
    if lo_ok and hi_ok:
        sos = scipy.signal.butter(order, [low, high], btype="band", output="sos")
    elif lo_ok:
        sos = scipy.signal.butter(order, low, btype="high", output="sos")
    elif hi_ok:
        sos = scipy.signal.butter(order, high, btype="low", output="sos")
    else:
        return data
    return scipy.signal.sosfiltfilt(sos, data)


def moving_average(a, n=3):
    """F2  util/filters.py:27-30."""
    c = np.cumsum(a, dtype=float)
    c[n:] = c[n:] - c[:-n]
    return c[n - 1:] / n


# ------------------------------------------------------------------------ trackers

def _interp_nans(y):
    """util/wow_detection.py:14-22."""
    bad = np.isnan(y)
    if bad.any():
        y[bad] = np.interp(bad.nonzero()[0], (~bad).nonzero()[0], y[~bad])


class _TrackGeometry:
    """W1  util/wow_detection.py:28-117 -- trail sampling and band -> bin arithmetic."""

    def __init__(self, spectrum, trail, fft_size, hop, sr, tolerance_st):
        self.spectrum = spectrum
        self.fft_size, self.hop, self.sr = fft_size, hop, sr
        self.num_bins, n_frames = spectrum.shape
        trail = sorted(trail, key=lambda t: t[0])
        t_raw = [t[0] for t in trail]
        f_raw = [t[1] for t in trail]
        self.frame_0, self.frame_1 = 0, n_frames
        if t_raw[0]:
            self.frame_0 = max(self.frame_0, int(t_raw[0] * sr / hop))
        if t_raw[-1]:
            self.frame_1 = min(self.frame_1, int(t_raw[-1] * sr / hop))
        self.times = np.linspace(self.frame_0 * hop / sr, self.frame_1 * hop / sr,
                                 self.frame_1 - self.frame_0)
        self.freqs = np.interp(self.times, t_raw, f_raw)
        self.tol = tolerance_st / 12
        self.NL = self.NU = 0

    def f2b(self, f):
        return max(1, min(self.num_bins - 1, int(round(f * self.fft_size / self.sr))))

    def band(self, freq, tol=None):
        tol = self.tol if tol is None else tol
        lf = np.log2(freq)
        return np.power(2, lf - tol), np.power(2, lf + tol)

    def limits(self, fL, fU):
        fL = max(1.0, fL)
        fU = min(self.sr / 2, fU)
        self.NL, self.NU = self.f2b(fL), self.f2b(fU)
        while self.NU - self.NL < 4:
            self.NL -= 1
            self.NU += 1

    def peak(self, i):
        """util/wow_detection.py:119-139 (allow_window is never True in shipped code)."""
        col = self.spectrum[:, self.frame_0 + i]
        # the all-ones window is kept: a band widened past the last bin makes this product raise (ValueError)
        # b stays a numpy int64 as in the reference: with float32 magnitudes (torch / pyfftw backends) the parabolic
        # offset is then float32 arithmetic but "+ x" promotes to float64 (a Python int would keep float32)
        b = self.NL + np.argmax(col[self.NL:self.NU] * np.ones(self.NU - self.NL))
        if col[b - 1] < col[b] > col[b + 1]:
            b, _ = parabolic(col, b)
        return b / self.fft_size * self.sr


def track_peak(spectrum, trail, fft_size, hop, sr, tolerance_st=1):
    """W2 PeakTracker  util/wow_detection.py:294-304."""
    g = _TrackGeometry(spectrum, trail, fft_size, hop, sr, tolerance_st)
    for i in range(len(g.freqs)):
        g.limits(*g.band(g.freqs[i]))
        g.freqs[i] = g.peak(i)
    _interp_nans(g.freqs)
    return g.times, g.freqs


def track_peak_track(spectrum, trail, fft_size, hop, sr, tolerance_st=1):
    """W2 PeakTrackTracker  util/wow_detection.py:307-327 (band stays on freqs[0];
    tolerance halves from the 4th frame on)."""
    g = _TrackGeometry(spectrum, trail, fft_size, hop, sr, tolerance_st)
    f0 = g.freqs[0]
    for i in range(len(g.freqs)):
        g.limits(*g.band(f0, g.tol / 2 if i > 2 else g.tol))
        g.freqs[i] = g.peak(i)
    _interp_nans(g.freqs)
    return g.times, g.freqs


def track_cog(spectrum, trail, fft_size, hop, sr, tolerance_st=1):
    """W2 CenterOfGravity  util/wow_detection.py:256-291."""
    g = _TrackGeometry(spectrum, trail, fft_size, hop, sr, tolerance_st)
    ff = fft_freqs(fft_size, sr)
    g.limits(*g.band(g.freqs[0]))
    for i in range(len(g.freqs)):
        w = np.hanning(g.NU - g.NL) * spectrum[g.NL:g.NU, g.frame_0 + i]
        g.freqs[i] = 2 ** (np.sum(w * np.log2(ff[g.NL:g.NU])) / np.sum(w))
        g.limits(*g.band(g.freqs[i]))
    _interp_nans(g.freqs)
    return g.times, g.freqs


def track_freehand(spectrum, trail, fft_size, hop, sr, tolerance_st=1):
    """W2 FreehandTracker  util/wow_detection.py:390-394."""
    g = _TrackGeometry(spectrum, trail, fft_size, hop, sr, tolerance_st)
    return g.times, g.freqs


def zero_crossings(a):
    """util/wow_detection.py:448-450."""
    pos = a > 0
    return np.where(np.bitwise_xor(pos[1:], pos[:-1]))[0]


def track_zero_crossing(spectrum, signal, trail, fft_size, hop, sr, tolerance_st=1):
    """W3 ZeroCrossingTracker  util/wow_detection.py:330-358 (signal is (n, ch))."""
    g = _TrackGeometry(spectrum, trail, fft_size, hop, sr, tolerance_st)
    fL, _ = g.band(np.min(g.freqs))
    _, fU = g.band(np.max(g.freqs))
    s0, s1 = int(g.times[0] * sr), int(g.times[-1] * sr)
    filt = butter_bandpass_filter(signal[s0:s1, 0], fL, fU, sr, order=3)
    cr = zero_crossings(filt)
    d = np.diff(cr).astype(np.float32)
    size = int(sr / 100 / np.mean(d))
    padded = np.pad(d, size, mode="reflect")
    w = scipy.signal.get_window("hann", size)
    dc = np.convolve(padded, w / size * 2, mode="same")[size:-size]
    g.freqs[:] = np.interp(g.times, cr[:len(dc)] / sr + g.times[0], sr / 2 / dc)
    _interp_nans(g.freqs)
    return g.times, g.freqs


def track_correlation(spectrum, trail, fft_size, hop, sr, tolerance_st=1):
    """W4 CorrelationTracker  util/wow_detection.py:396-436 (reads columns 0.. -- the
    reference ignores frame_0 here, quirk kept)."""
    g = _TrackGeometry(spectrum, trail, fft_size, hop, sr, tolerance_st)
    ff = fft_freqs(fft_size, sr)
    fL, fU = min(g.freqs), max(g.freqs)
    g.limits(fL, fU)
    ns = (g.NU - g.NL) * 4
    lf = np.log2(ff[g.NL:g.NU])
    grid = np.linspace(lf[0], lf[-1], ns)
    nf = len(g.freqs)
    res = np.ones((ns, nf + 1))
    for i in range(nf):
        res[:, i] = scipy.interpolate.interp1d(lf, spectrum[g.NL:g.NU, i], kind="quadratic")(grid)
    wind = np.hanning(ns)
    changes = np.ones(nf)
    for i in range(nf):
        r = xcorr(res[:, i] * wind, res[:, i + 1] * wind, mode="same")
        ip, _ = parabolic(r, int(np.argmax(r)))
        changes[i] = (ns // 2) - ip
    speed = np.cumsum(changes) / ns * (lf[-1] - lf[0])
    g.freqs[:] = np.power(2, np.log2((fL + fU) / 2) + speed)
    _interp_nans(g.freqs)
    return g.times, g.freqs


TRACKERS = {
    "Peak": track_peak,
    "Peak Track": track_peak_track,
    "Center of Gravity": track_cog,
    "Freehand Draw": track_freehand,
    "Correlation": track_correlation,
}


# ------------------------------------------------- P0: headless pyrespeeder data flow

def piptrack(S, sr, n_fft, fmin=150.0, fmax=4000.0, threshold=0.1):
    """librosa.piptrack(S=...) restated from librosa 0.10's published source (core/pitch.py): the routine PartialsTracker
    calls (util/wow_detection.py:361-387).  librosa is a third-party dependency that is neither vendored in the reference
    nor installed here (requirements.txt names it without a version): PARITY UNPINNED -- no golden vector exists; the
    restatement is checked by hand-computed cases in tests/test_oracle_golden.py.  S: magnitudes (bins, frames)."""
    S = np.abs(np.asarray(S))
    fmin = np.maximum(fmin, 0)
    fmax = np.minimum(fmax, float(sr) / 2)
    fft_freqs = np.fft.rfftfreq(n_fft, 1.0 / sr)
    avg = np.gradient(S, axis=-2)
    a = S[2:] + S[:-2] - 2 * S[1:-1]                       # _parabolic_interpolation along the frequency axis
    b = (S[2:] - S[:-2]) / 2
    shift = np.zeros_like(S)
    with np.errstate(divide="ignore", invalid="ignore"):
        inner = np.where(np.abs(b) < np.abs(a), -b / a, 0)
    shift[1:-1] = inner
    dskew = 0.5 * avg * shift
    pitches, mags = np.zeros_like(S), np.zeros_like(S)
    freq_mask = ((fmin <= fft_freqs) & (fft_freqs < fmax))[:, None]
    ref_value = threshold * np.max(S, axis=-2)[None, :]
    x = S * (S > ref_value)
    xp = np.pad(x, ((1, 1), (0, 0)), mode="edge")          # util.localmax: strictly above the left, at least the right neighbour
    idx = np.nonzero(freq_mask & (x > xp[:-2]) & (x >= xp[2:]))
    pitches[idx] = (idx[0] + shift[idx]) * float(sr) / n_fft
    mags[idx] = S[idx] + dskew[idx]
    return pitches, mags


def trace_to_speed(freqs):
    """util/markers.py:197-199 (TraceLine, offset 0): log2 speed centred on 0."""
    s = np.log2(freqs)
    return s - np.mean(s)


def master_speed_curve(lines, duration, sr, hop, bands=(0, 20)):
    """P0  util/markers.py:585-639 (MasterSpeedLine.update + get_linspace).

    ``lines`` is a list of (times, log2speed).  Returns the (num, 2) array
    [[t_seconds, linear speed], ...] handed to resampling.run.
    Note out[] is float32 inside sample_lines (util/markers.py:609).
    """
    marker_sr = sr / hop
    num = int(duration * marker_sr)
    times = np.linspace(0, duration, num=num)
    cols = np.zeros((len(times), len(lines)), dtype=np.float32)
    for i, (lt, lv) in enumerate(lines):
        cols[:, i] = np.interp(times, lt, lv, left=np.nan, right=np.nan)
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mean = np.nanmean(cols, axis=1)
    _interp_nans(mean)
    lo, hi = sorted(bands)
    filt = butter_bandpass_filter(mean, lo, hi, marker_sr, order=3)
    data = np.stack((times, filt), axis=-1)
    out = np.array(data)
    np.power(2, out[:, 1], out[:, 1])
    return out


# ------------------------------------------------------------- config 4: dropout healer

def master_reg_curve(regs, duration, sr, hop):
    """MasterRegLine.update + BaseLine.get_linspace (util/markers.py:670-708, 593-597) on RegLine objects as
    RegLine.__init__ leaves them (:94-119: negative amplitudes folded into the phase).  regs: to_cfg rows."""
    marker_sr = sr / hop
    times = np.linspace(0, duration, num=int(duration * marker_sr))
    objs = []
    for t0, t1, amplitude, omega, phase, offset in regs:
        if amplitude < 0:
            amplitude *= -1
            phase += np.pi
        objs.append(dict(t_center=(t0 + t1) / 2, amplitude=amplitude, omega=omega, phase=phase, offset=offset))
    objs.sort(key=lambda r: r["t_center"])
    pi2 = 2 * np.pi
    t_centers, amp_centers, phi_centers = [], [], []
    for i, reg in enumerate(objs):
        if i == 0:
            phi_centers.append(reg["omega"] * times[0] + reg["phase"] % pi2 + reg["offset"] * pi2)
            t_centers.append(times[0])
            amp_centers.append(reg["amplitude"])
        phi_centers.append(reg["omega"] * reg["t_center"] + reg["phase"] % pi2 + reg["offset"] * pi2)
        t_centers.append(reg["t_center"])
        amp_centers.append(reg["amplitude"])
        if i == len(objs) - 1:
            phi_centers.append(reg["omega"] * times[-1] + reg["phase"] % pi2 + reg["offset"] * pi2)
            t_centers.append(times[-1])
            amp_centers.append(reg["amplitude"])
    out = np.stack((times, 1.5 * np.interp(times, t_centers, amp_centers) * np.sin(np.interp(times, t_centers, phi_centers))), axis=-1)
    np.power(2, out[:, 1], out[:, 1])
    return out


def project_speed_curve(cfg, duration, sr):
    """pyrespeeder_gui.Canvas.get_speed_curve (pyrespeeder_gui.py:133-140) for the markers of a .spd project
    (util/widgets.py:1247-1262): TraceLine.__init__'s log2 speed + offset (util/markers.py:196-214) per line."""
    hop = cfg["fft_size"] // cfg.get("fft_overlap", 1)
    if cfg.get("regs"):
        return master_reg_curve(cfg["regs"], duration, sr, hop)
    lines = []
    for times, freqs, offset in cfg["lines"]:
        speed = np.log2(np.asarray(freqs, dtype=np.float64))
        speed -= np.mean(speed)
        lines.append((np.asarray(times, dtype=np.float64), speed + (0 if offset is None else offset)))
    return master_speed_curve(lines, duration, sr, hop, (cfg.get("highpass", 0), cfg.get("lowpass", 20)))


def heal_dropouts(signal, sr, markers, fft_size=512, hop=32, stft_fn=None, istft_fn=None):
    """dropout_healer_gui.Canvas.resample_files (dropout_healer_gui.py:111-166), one pass per channel.
    markers: (a0, a1, b0, b1, surrounding) as DropoutSample.to_cfg() (util/markers.py:368-426).
    stft_fn/istft_fn default to this module's restatements (gen_golden passes the reference's own)."""
    from scipy.interpolate import RegularGridInterpolator
    stft_fn = stft_fn or stft
    istft_fn = istft_fn or istft
    sig2d = signal[:, None] if signal.ndim == 1 else signal
    n, ch = sig2d.shape
    out = np.empty(sig2d.shape, dtype=sig2d.dtype)
    y_pad = fix_length(sig2d, n + fft_size // 2)

    def t2f(t):
        return int(t * sr / hop)

    def f2b(f):
        return max(1, min(fft_size // 2, int(round(f * fft_size / sr))))

    for c in range(ch):
        S = np.array(stft_fn(y_pad[:, c], n_fft=fft_size, step=hop))
        db = 20 * np.log10(np.abs(S) + .0000001)
        gain_whole = np.zeros(S.shape, dtype=float)
        for (a0, a1, b0, b1, surrounding) in markers:
            width, t = abs(a0 - b0), (a0 + b0) / 2
            f, height = (a1 + b1) / 2, abs(a1 - b1)
            frame_b, frame_a = t2f(t - width / 2), t2f(t + width / 2)
            fs = max(1, t2f(width * surrounding))
            bin_l, bin_u = f2b(f - height / 2), f2b(f + height / 2)
            before = np.mean(db[bin_l:bin_u, frame_b - fs:frame_b], axis=1)
            after = np.mean(db[bin_l:bin_u, frame_a:frame_a + fs], axis=1)
            fp_frames = np.linspace(frame_b, frame_a, num=frame_a - frame_b)
            fp_bins = np.linspace(bin_l, bin_u, num=bin_u - bin_l)
            interp = RegularGridInterpolator(((frame_b, frame_a), fp_bins), (before, after))
            mp_bins, mp_frames = np.meshgrid(fp_bins, fp_frames)
            fp_db = np.swapaxes(interp((mp_frames, mp_bins)), 0, 1)
            g = fp_db - db[bin_l:bin_u, frame_b:frame_a]
            np.clip(g, gain_whole[bin_l:bin_u, frame_b:frame_a], 255, out=g)
            gain_whole[bin_l:bin_u, frame_b:frame_a] = g
        S = S * np.power(10, gain_whole / 20)
        out[:, c] = istft_fn(S, length=n, hop_length=hop)
    return out


def band_volume_db(imdata, sr, fft_size, hop, t_0, t_1, f_lower, f_upper):
    """dropout_healer_gui.py:195-203: mean over the band of to_dB(magnitude), per frame."""
    db = 20 * np.log10(np.array(imdata))
    frame_b, frame_a = int(t_0 * sr / hop), int(t_1 * sr / hop)
    f2b = lambda f: max(1, min(fft_size // 2, int(round(f * fft_size / sr))))
    return np.mean(db[f2b(f_lower):f2b(f_upper), frame_b:frame_a], axis=0), frame_b


def detect_dropouts(imdata, sr, fft_size, hop, t_0, t_1, f_lower, f_upper, width_ms=20, sensitivity=5):
    """Batch detection branch of dropout_healer_gui.Canvas.on_mouse_release (dropout_healer_gui.py:185-242).
    imdata = the cached magnitude spectrogram (bins, frames).  -> list of corner pairs (:240)."""
    import scipy.signal
    from scipy.signal import savgol_filter
    vol, frame_b = band_volume_db(imdata, sr, fft_size, hop, t_0, t_1, f_lower, f_upper)
    t2f = lambda t: int(t * sr / hop)
    f2t = lambda f: f / sr * hop
    half_width = width_ms / 1000 / 2
    fhw = t2f(half_width)
    vol_lt = savgol_filter(vol, fhw * 12, 5)
    vol_st = savgol_filter(vol, fhw, 5)
    peaks, _ = scipy.signal.find_peaks(-vol, height=None, threshold=None, distance=None, prominence=10.0 - sensitivity,
                                       wlen=None, rel_height=0.5, plateau_size=None)
    found = []
    for f_peak in peaks:
        t_center = f2t(frame_b + f_peak)
        try:
            f_qw = t2f(half_width / 4)
            f_before, f_after = f_peak - f_qw, f_peak + f_qw
            xp = np.arange(f_before, f_after)
            parabola = np.poly1d(np.polyfit(xp, vol_st[f_before:f_after], 2), r=False, variable=None)
            f_hw = t2f(half_width)
            f_before, f_after = f_peak - f_hw, f_peak + f_hw
            xp = np.arange(f_before, f_after)
            fp = parabola(xp)
            f_intersection = scipy.signal.argrelmin(np.abs(fp - vol_lt[f_before:f_after]))[0]
            assert len(f_intersection) == 2
            half_width = f2t(f_intersection[1] - f_intersection[0])
        except Exception:
            pass
        found.append(((t_center - half_width, f_lower), (t_center + half_width, f_upper)))
    return found


# ------------------------------------------------------------------------- heuristic dropout repair
def heuristic_bands(f_lower, f_upper, num_bands):
    """Band edges of dropouts_gui.py:252: a log-spaced grid cast to uint16 (the edges stay numpy uint16 scalars: what follows
    from that is part of the reference's behaviour, see heuristic_bins)."""
    return np.logspace(np.log2(f_lower), np.log2(f_upper), num=num_bands, endpoint=True, base=2, dtype=np.uint16)


def heuristic_bins(f_lower_band, f_upper_band, fft_size, sr):
    """dropouts_gui.py:281-282: int(f * fft_size / sr) with f a numpy uint16 scalar.  Under the numpy this build is pinned to
    (2.x, NEP 50) uint16 * python int stays uint16 and WRAPS for f * fft_size >= 65536 (3000 Hz * 512 -> 28672): the reference's
    bands then sit in the lowest bins whatever their frequencies.  Evaluated through numpy itself, so this restatement follows
    the installed numpy exactly as the reference does."""
    with np.errstate(over="ignore"):
        return int(f_lower_band * fft_size / sr), int(f_upper_band * fft_size / sr)


def heuristic_gain_curve(vol, d, max_slope):
    """dropouts_gui.py:287-311 for one band: valleys of the volume curve (scipy find_peaks on -vol, prominence 5 dB), each patched
    by the straight line between the mean volumes d..2d frames to its left and right when that line's slope stays below
    max_slope.  Returns the gain curve in dB (zeros elsewhere)."""
    import scipy.signal
    n_frames = len(vol)
    peaks, _ = scipy.signal.find_peaks(-vol, height=None, threshold=None, distance=None, prominence=5, wlen=None,
                                       rel_height=0.5, plateau_size=None)
    gain_curve = np.zeros(n_frames)
    for peak_i in peaks:
        if 2 * d < peak_i < n_frames - 2 * d - 1:
            left = np.mean(vol[peak_i - 2 * d:peak_i - d])
            right = np.mean(vol[peak_i + d:peak_i + 2 * d])
            m = (left - right) / (2 * d)
            if abs(m) < max_slope:
                gain_curve[peak_i - d:peak_i + d + 1] = np.interp(range(2 * d + 1), (0, 2 * d), (left, right)) - vol[peak_i - d:peak_i + d + 1]
    return gain_curve


def heal_heuristic(signal, sr, fft_size, hop, max_width=0.02, max_slope=0.5, num_bands=3, bottom_freedom=2, f_upper=12000,
                   f_lower=3000):
    """dropouts_gui.MainWindow.process_heuristic (dropouts_gui.py:241-323) for one file: signal (n, ch) float32 -> repaired
    copy.  Per channel: dB spectrogram (hann), then from the top band down -- band volume, valley gains, the factor clipped
    between 1 and bottom_freedom x the band above's, the signal x (factor - 1) band-passed (order 3, zero phase) and ADDED to
    the signal the next band then starts from (:314-321: the bands are sequential through the signal)."""
    signal = np.array(signal, dtype=np.float32)
    bands = heuristic_bands(f_lower, f_upper, num_bands)
    d = int(max_width / 1.5 * sr / hop)
    n = signal.shape[0]
    with np.errstate(all="ignore"):
        for channel in range(signal.shape[1]):
            imdata = np.array(20 * np.log10(get_mag(signal[:, channel], fft_size, hop, "hann")))
            correction_fac = np.ones(imdata.shape[1]) * 1000
            edges = list(zip(bands[:-1], bands[1:]))
            for f_lo, f_hi in reversed(edges):
                bin_lower, bin_upper = heuristic_bins(f_lo, f_hi, fft_size, sr)
                vol = np.mean(imdata[bin_lower:bin_upper], axis=0)
                gain_curve = heuristic_gain_curve(vol, d, max_slope)
                correction_fac = np.clip(np.power(10, gain_curve / 20), 1, correction_fac * bottom_freedom)
                vol_corr = signal[:, channel] * np.interp(np.linspace(0, 1, n), np.linspace(0, 1, len(correction_fac)),
                                                          correction_fac - 1)
                signal[:, channel] += butter_bandpass_filter(vol_corr, f_lo, f_hi, sr, order=3)
    return signal
