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


def _padlen(sos):
    ntaps = 2 * sos.shape[0] + 1
    ntaps -= min((sos[:, 2] == 0).sum(), (sos[:, 5] == 0).sum())
    return int(ntaps * 3)


def sosfiltfilt_batch_dev(sos, x_t, dev=None):
    """scipy.signal.sosfiltfilt along the last axis of a 2-D float64 device tensor (n_sig, n) -- every stage one launch over the
    whole batch (K_sosfiltfilt's batched form).  sos: one cascade (n_sections, 6) for all rows, or a sequence of n_sig cascades
    of equal section count and padding (e.g. the bands of dropouts_gui.process_heuristic).  Each row equals sosfiltfilt_dev's."""
    dev = _dev.device_index(dev if dev is not None else x_t.device)
    if x_t.ndim != 2:
        raise ValueError("x must be 2D (n_sig, n)")
    x_t = x_t.contiguous()
    n_sig, n = x_t.shape
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    if sos.ndim == 2:
        sos = sos[None]
    if sos.ndim != 3 or sos.shape[2] != 6 or sos.shape[0] not in (1, n_sig):
        raise ValueError(f"sos must be (n_sections, 6) or ({n_sig}, n_sections, 6), got {sos.shape}")
    pads = {_padlen(c) for c in sos}
    if len(pads) != 1:
        raise ValueError("the cascades of one batch must share scipy's default padlen")
    padlen = pads.pop()
    if n <= padlen:
        raise ValueError("The length of the input vector x must be greater than padlen, which is %d." % padlen)
    zi = np.ascontiguousarray([scipy.signal.sosfilt_zi(c) for c in sos], dtype=np.float64)
    L = _lib.lib()
    work = _dev.empty(int(L.par_sosfiltfilt_batch_work_len(n, padlen, n_sig, sos.shape[1])), torch.float64, dev)
    y = _dev.empty((n_sig, n), torch.float64, dev)
    _lib.check(L.par_sosfiltfilt_batch_f64(dev, sos.ctypes.data_as(ctypes.c_void_p), zi.ctypes.data_as(ctypes.c_void_p),
                                           sos.shape[0], sos.shape[1], _dev.ptr(x_t), n, n_sig, n, padlen, _dev.ptr(work),
                                           work.numel(), _dev.ptr(y), n, _dev.stream_ptr(dev)))
    return y


def bandpass_batch_dev(data_t, lowcuts, highcuts, fs, order=5, dev=None):
    """butter_bandpass_filter of every row of a (n_sig, n) float64 device tensor with its own band (scalars: one band for all).
    Every band must have both cut-offs inside (0, Nyquist) or every band the same one side (the cascades share their shape)."""
    n_sig = data_t.shape[0]
    lows = np.broadcast_to(np.asarray(lowcuts, dtype=np.float64), (n_sig,)) if np.ndim(lowcuts) else [lowcuts]
    highs = np.broadcast_to(np.asarray(highcuts, dtype=np.float64), (n_sig,)) if np.ndim(highcuts) else [highcuts]
    if len(lows) == 1 and len(highs) == 1:
        sos = _design(float(lows[0]), float(highs[0]), fs, order)
        return data_t if sos is None else sosfiltfilt_batch_dev(sos, data_t, dev)
    lows, highs = np.broadcast_to(lows, (n_sig,)), np.broadcast_to(highs, (n_sig,))
    designs = [_design(float(a), float(b), fs, order) for a, b in zip(lows, highs)]
    if any(d is None for d in designs):
        raise ValueError("a band of the batch has no cut-off inside (0, Nyquist)")
    return sosfiltfilt_batch_dev(np.stack(designs), data_t, dev)


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
    return y if isinstance(data, torch.Tensor) else _dev.to_host(y)


def moving_average(a, n=3):
    """Box filter over n samples, len(a) - n + 1 outputs (util/filters.py:27-30)."""
    return np.convolve(np.asarray(a, dtype=float), np.full(n, 1.0 / n), mode="valid")


def make_odd(n):
    """n, or the next odd number (util/filters.py:33-36)."""
    return n if n % 2 else n + 1
