"""Correlation helpers with the call contracts of the reference's util/correlation.py (xcorr :6-13,
find_delay :16-39, parabolic :42-46), on the device: `par_xcorr_f64` / `par_find_delay_f64` (csrc/stft.hip) run the
correlation through one complex four-step FFT of a + i b; find_delay re-evaluates the lags around the peak as
float64 dot products before the parabolic refinement, so the delay the tape-sync tool reads is exact.
"""
import ctypes
import logging

import numpy as np
import torch
from scipy.signal import get_window

from . import _dev, _lib


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


def _up(a, dev):
    return a.to(device=f"cuda:{dev}", dtype=torch.float64) if isinstance(a, torch.Tensor) else \
        _dev.to_dev(a, torch.float64, dev)                  # float32 signals cross PCIe as they are and widen in HBM


def xcorr_dev(a_t, b_t, dev=None):
    """'full' normalised cross-correlation of two 1-D float64 device tensors -> float64 device tensor."""
    dev = _dev.device_index(dev if dev is not None else a_t.device)
    L = _lib.lib()
    na, nb = a_t.numel(), b_t.numel()
    nbytes = int(L.par_xcorr_scratch_bytes(na, nb))
    scratch = _dev.empty(max(nbytes, 1), torch.uint8, dev)
    full = _dev.empty(na + nb - 1, torch.float64, dev)
    _lib.check(L.par_xcorr_f64(dev, _dev.ptr(a_t), na, _dev.ptr(b_t), nb, _dev.ptr(scratch), nbytes, _dev.ptr(full),
                               _dev.stream_ptr(dev)))
    return full


def xcorr(a, b, mode='full'):
    """Normalised cross-correlation in [-1, 1] of two 1-D signals (scipy.signal.correlate conventions)."""
    dev = _dev.device_index(None)
    a, b = np.asarray(a), np.asarray(b)
    if a.ndim != 1 or b.ndim != 1 or len(a) < 1 or len(b) < 1:
        raise ValueError("xcorr needs two non-empty 1-D signals")
    full = xcorr_dev(_up(a, dev), _up(b, dev), dev).cpu().numpy()
    return _centre(full, len(a), len(b), mode)


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
        if isinstance(sig, torch.Tensor):          # device-resident signals (pipeline.correlate_sources): windowed in HBM
            sig.mul_(torch.from_numpy(get_window(window_name, sig.numel())).to(device=sig.device, dtype=sig.dtype))
        else:
            sig *= get_window(window_name, len(sig))
    if ignore_phase:
        logging.warning("Ignoring phase")
    dev = _dev.device_index(None)
    L = _lib.lib()
    a_t, b_t = _up(a, dev), _up(b, dev)
    na, nb = a_t.numel(), b_t.numel()
    nbytes = int(L.par_xcorr_scratch_bytes(na, nb))
    scratch = _dev.empty(max(nbytes, 1), torch.uint8, dev)
    delay, corr = ctypes.c_double(0.0), ctypes.c_double(0.0)
    _lib.check(L.par_find_delay_f64(dev, _dev.ptr(a_t), na, _dev.ptr(b_t), nb, int(bool(ignore_phase)), _dev.ptr(scratch), nbytes,
                                    ctypes.byref(delay), ctypes.byref(corr), _dev.stream_ptr(dev)))
    logging.debug(f"i_peak {delay.value + na // 2}")
    return delay.value, corr.value
