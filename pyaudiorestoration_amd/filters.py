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
    """SOS Butterworth design for the cut-offs that lie strictly inside (0, Nyquist): both -> band-pass, only the
    lower -> high-pass, only the upper -> low-pass, neither -> None (util/filters.py:7-23)."""
    edges = {"low": lowcut / (0.5 * fs), "high": highcut / (0.5 * fs)}
    usable = {k: v for k, v in edges.items() if 0 < v < 1}
    if len(usable) == 2:
        return scipy.signal.butter(order, [usable["low"], usable["high"]], btype='band', output='sos')
    if "low" in usable:
        return scipy.signal.butter(order, usable["low"], btype='high', output='sos')
    if "high" in usable:
        return scipy.signal.butter(order, usable["high"], btype='low', output='sos')
    return None


def bandpass_dev(data, lowcut, highcut, fs, order=5, dev=None):
    """butter_bandpass_filter whose result stays in HBM (float64 device tensor) for a next device stage;
    numpy or tensor input.  Neither cut-off inside (0, Nyquist): the input itself, uploaded if needed."""
    dev = _dev.device_index(dev if dev is not None else (data.device if isinstance(data, torch.Tensor) else None))
    x_t = _dev.to_dev(data if isinstance(data, torch.Tensor) else np.asarray(data), torch.float64, dev)
    sos = _design(lowcut, highcut, fs, order)
    return x_t if sos is None else sosfiltfilt_dev(sos, x_t, dev)


def butter_bandpass_filter(data, lowcut, highcut, fs, order=5):
    """Zero-phase Butterworth band / high / low-pass, or the input itself when neither cut-off lies inside
    (0, Nyquist) -- the contract of the reference's util/filters.py:7-24.  The design is scipy's (host, a few
    coefficients); the filtering runs in K_sosfiltfilt.  numpy in -> numpy out, device tensor in -> device tensor."""
    if _design(lowcut, highcut, fs, order) is None:
        return data
    y = bandpass_dev(data, lowcut, highcut, fs, order)
    return y if isinstance(data, torch.Tensor) else y.cpu().numpy()


def moving_average(a, n=3):
    """Box filter over n samples, len(a) - n + 1 outputs (util/filters.py:27-30)."""
    return np.convolve(np.asarray(a, dtype=float), np.full(n, 1.0 / n), mode="valid")


def make_odd(n):
    """n, or the next odd number (util/filters.py:33-36)."""
    return n if n % 2 else n + 1
