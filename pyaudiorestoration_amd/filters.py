"""Mirror of reference util/filters.py: butter_bandpass_filter :7-24, moving_average :27-30,
make_odd :33-37.  Filter design (O(order)) is scipy on the host, like the reference; the O(n)
zero-phase filtering runs in K_sosfiltfilt (csrc/filt.hip) instead of scipy's C loop."""
import ctypes

import numpy as np
import scipy.signal
import torch

from . import _dev, _lib


def sosfiltfilt_dev(sos, x_t, dev=None):
    """scipy.signal.sosfiltfilt(sos, x) for a 1-D float64 device tensor (default padding)."""
    dev = _dev.device_index(dev if dev is not None else x_t.device)
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    n_sections = sos.shape[0]
    ntaps = 2 * n_sections + 1
    ntaps -= min((sos[:, 2] == 0).sum(), (sos[:, 5] == 0).sum())
    padlen = int(ntaps * 3)
    n = x_t.numel()
    if n <= padlen:
        raise ValueError("The length of the input vector x must be greater than padlen, which is %d." % padlen)
    zi = np.ascontiguousarray(scipy.signal.sosfilt_zi(sos), dtype=np.float64)
    L = _lib.lib()
    work = _dev.empty(int(L.par_sosfiltfilt_work_len(n, padlen)), torch.float64, dev)
    y = _dev.empty(n, torch.float64, dev)
    _lib.check(L.par_sosfiltfilt_f64(dev, sos.ctypes.data_as(ctypes.c_void_p), zi.ctypes.data_as(ctypes.c_void_p),
                                     n_sections, _dev.ptr(x_t), n, padlen, _dev.ptr(work), work.numel(), _dev.ptr(y),
                                     _dev.stream_ptr(dev)))
    return y


def _design(lowcut, highcut, fs, order):
    nyq = 0.5 * fs
    low = lowcut / nyq
    high = highcut / nyq
    low_in_range = 0 < low < 1
    high_in_range = 0 < high < 1
    if low_in_range and high_in_range:
        return scipy.signal.butter(order, [low, high], btype='band', output='sos')
    elif low_in_range and not high_in_range:
        return scipy.signal.butter(order, low, btype='high', output='sos')
    elif not low_in_range and high_in_range:
        return scipy.signal.butter(order, high, btype='low', output='sos')
    return None


def butter_bandpass_filter(data, lowcut, highcut, fs, order=5):
    """Performs a low, high or bandpass filter if low & highcut are in range"""
    sos = _design(lowcut, highcut, fs, order)
    if sos is None:
        return data
    if isinstance(data, torch.Tensor):
        return sosfiltfilt_dev(sos, data.to(torch.float64))
    dev = _dev.device_index(None)
    x_t = _dev.to_dev(np.asarray(data), torch.float64, dev)
    return sosfiltfilt_dev(sos, x_t, dev).cpu().numpy()


def moving_average(a, n=3):
    ret = np.cumsum(a, dtype=float)
    ret[n:] = ret[n:] - ret[:-n]
    return ret[n - 1:] / n


def make_odd(n):
    return n if n % 2 else n + 1
