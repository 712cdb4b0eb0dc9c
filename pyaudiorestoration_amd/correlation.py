"""Correlation helpers with the call contracts of the reference's util/correlation.py (xcorr :6-13,
find_delay :16-39, parabolic :42-46).  Host code: they only ever see window-sized vectors (CorrelationTracker,
tape-sync markers); the implementation is this package's own (FFT correlation, batched form for the tracker).
"""
import logging

import numpy as np
from scipy.signal import get_window


def _next_pow2(n):
    return 1 << max(0, int(n) - 1).bit_length()


def xcorr_rows(A, B):
    """L2-normalised 'full' cross-correlation of matching rows of A and B (shape (rows, N) and (rows, M)):
    out[r, j] = sum_n a[r, n + j - (M-1)] * b[r, n], j = 0 .. N+M-2, computed with one batched real FFT."""
    A = np.atleast_2d(np.asarray(A, dtype=np.float64))
    B = np.atleast_2d(np.asarray(B, dtype=np.float64))
    A = A / np.sqrt(np.einsum("ij,ij->i", A, A))[:, None]
    B = B / np.sqrt(np.einsum("ij,ij->i", B, B))[:, None]
    n, m = A.shape[1], B.shape[1]
    nfft = _next_pow2(n + m - 1)
    circ = np.fft.irfft(np.fft.rfft(A, nfft, axis=1) * np.conj(np.fft.rfft(B, nfft, axis=1)), nfft, axis=1)
    # lag l = j - (M-1): negative lags wrap to the end of the circular result
    return np.concatenate((circ[:, nfft - (m - 1):] if m > 1 else circ[:, :0], circ[:, :n]), axis=1)


def _centre(full, n, m, mode):
    if mode == "full":
        return full
    if mode == "same":                       # size of the first input, centred on the 'full' output (scipy)
        start = (m - 1) // 2
        return full[..., start:start + n]
    if mode == "valid":
        lo, hi = min(n, m), max(n, m)
        return full[..., lo - 1:hi]
    raise ValueError(f"unknown correlation mode {mode!r}")


def xcorr(a, b, mode='full'):
    """Normalised cross-correlation in [-1, 1] of two 1-D signals (scipy.signal.correlate conventions)."""
    a = np.asarray(a)
    b = np.asarray(b)
    return _centre(xcorr_rows(a[None, :], b[None, :])[0], len(a), len(b), mode)


def parabolic(f, x):
    """Vertex (position, height) of the parabola through f[x-1], f[x], f[x+1]."""
    left, mid, right = f[x - 1], f[x], f[x + 1]
    slope2 = left - right                                  # twice the centred difference
    xv = x + 0.5 * slope2 / (left - 2 * mid + right)
    return xv, mid - 0.25 * slope2 * (xv - x)


def find_delay(a, b, ignore_phase=False, window_name=None):
    """Delay in samples (fractional) between 1-D signals a and b, and the correlation there.
    Like the reference, a given window is applied to `a` and `b` IN PLACE."""
    for sig in ((a, b) if window_name else ()):
        sig *= get_window(window_name, len(sig))
    res = xcorr(a, b, mode="same")
    if ignore_phase:
        logging.warning("Ignoring phase")
    best = int(np.argmax(np.abs(res) if ignore_phase else res))
    i_peak, corr = parabolic(res, best)
    logging.debug(f"i_peak {i_peak}")
    return i_peak - len(res) // 2, corr
