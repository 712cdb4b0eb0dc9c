"""Mirror of reference util/fourier.py -- same names, arguments and error behaviour, HIP inside.

stft :37-75, get_mag :27-29, to_mag :23-24, istft :314-437, fix_length :440-478,
fft_freqs :690-700.  `hip_rfft2` has the exact backend-slot signature
``(n_fft, step, window, x, zeropad)`` of util/fourier.py:67-70, so a maintainer can put it first in
that tuple.  Arrays are returned with the reference's logical shape
``(n_freqs, n_frames)``; memory is frame-major (a transposed view), which the reference's own numpy
backend also produces for n_fft > 512 (util/fourier.py:147).
"""
import logging
from time import perf_counter

import numpy as np
import torch
from scipy import signal as dsp

from . import _dev, _lib

# kept for API parity (util/fourier.py:20-21); the HIP ISTFT needs no column blocking
MAX_MEM_BLOCK = 2 ** 8 * 2 ** 10


def to_mag(spectrum):
    return abs(spectrum) + .0000001


def get_mag(*args, **kwargs):
    """Get the magnitude spectrum from complex input (fused: |X|+1e-7 never round-trips a complex
    spectrogram through HBM)."""
    return stft(*args, _mode=1, **kwargs)


class timed_log:
    """`with timed_log("hip"):` logs "<name> <seconds>s" at INFO on exit (the reference's backends do the same)."""

    def __init__(self, method_name):
        self.method_name = method_name

    def __enter__(self):
        self.t0 = perf_counter()

    def __exit__(self, *exc):
        logging.info("%s %0.2fs", self.method_name, perf_counter() - self.t0)


_window_cache = {}


def window_dev(window_name, n_fft, dev):
    """scipy.signal.get_window(name, n_fft) (periodic, like util/fourier.py:66) as a float32 device tensor, kept per
    (name, size, device): a pageable upload per call is a stream synchronisation per call."""
    key = (window_name if isinstance(window_name, (str, tuple)) else None, int(n_fft), dev)
    if key[0] is None:
        return _dev.to_dev(dsp.get_window(window_name, n_fft).astype(np.float32), torch.float32, dev)
    w = _window_cache.get(key)
    if w is None:
        if len(_window_cache) > 64:
            _window_cache.clear()
        w = _window_cache[key] = _dev.to_dev(dsp.get_window(window_name, n_fft).astype(np.float32), torch.float32, dev)
    return w


def stft_dev(x_t, n_fft, step, window_t, zeropad=1, mode=0, x_stride=1, n=None, dev=None):
    """Device-resident STFT.  x_t: float32 tensor (element stride x_stride, logical length n).
    Returns a tensor with logical shape (bins, frames) that is a transposed view of the frame-major
    [frames][bins] buffer the kernel wrote (complex64 for mode 0, float32 magnitude for mode 1)."""
    dev = _dev.device_index(dev if dev is not None else x_t.device)
    L = _lib.lib()
    if n is None:
        n = x_t.numel() // x_stride
    bins = (n_fft * zeropad) // 2 + 1
    frames = int(L.par_stft_frames(n, n_fft, step))
    if mode == 1 and n_fft * zeropad <= 16384 and bins >= 32:
        # magnitude rows start on 128-byte lines (packed 2052-byte rows of the usual 1024-point transform straddle lines at
        # both ends: 1.21x the bytes written); the caller gets the (bins, frames) view of the pitched buffer
        pitch = (bins + 31) // 32 * 32
        buf = _dev.empty((frames, pitch), torch.float32, dev)
        _lib.check(L.par_stft_f32(dev, _dev.ptr(x_t), n, x_stride, n_fft, step, zeropad, _dev.ptr(window_t), _dev.ptr(buf), mode,
                                  pitch, _dev.stream_ptr(dev)))
        return buf[:, :bins].T
    out = _dev.empty((frames, bins), torch.complex64 if mode == 0 else torch.float32, dev)
    if n_fft * zeropad > 16384:           # four-step transform through a scratch of one complex H-point array per frame of a batch
        nbytes = int(L.par_stft_big_scratch_bytes(n, n_fft, step, zeropad))
        scratch = _dev.empty(max(nbytes, 1), torch.uint8, dev)
        _lib.check(L.par_stft_big_f32(dev, _dev.ptr(x_t), n, x_stride, n_fft, step, zeropad, _dev.ptr(window_t), _dev.ptr(out),
                                      mode, _dev.ptr(scratch), nbytes, _dev.stream_ptr(dev)))
        return out.T
    _lib.check(L.par_stft_f32(dev, _dev.ptr(x_t), n, x_stride, n_fft, step, zeropad, _dev.ptr(window_t), _dev.ptr(out),
                              mode, 0, _dev.stream_ptr(dev)))
    return out.T


def hip_rfft2(n_fft, step, window, x, zeropad, _mode=0):
    """Backend-slot callable (util/fourier.py:67-70).  Raises on any failure so that the
    reference's chain falls through to its next backend."""
    with timed_log("hip"):
        dev = _dev.device_index(None)
        x_t = _dev.to_dev(x, torch.float32, dev)
        w_t = _dev.to_dev(window, torch.float32, dev)
        res = stft_dev(x_t, n_fft, step, w_t, zeropad, _mode, dev=dev)
        return _dev.to_host(res)


def stft(x, n_fft=1024, step=512, window_name='blackmanharris', zeropad=1, _mode=0):
    """Compute the STFT; returns ndarray of shape (n_freqs, n_steps) like the reference."""
    n_fft = int(n_fft)
    step = max(n_fft // 2, 1) if step is None else int(step)
    if isinstance(x, torch.Tensor):
        if x.ndim != 1:
            raise ValueError('x must be 1D')
        dev = _dev.device_index(x.device)
        window = window_dev(window_name, n_fft, dev)
        xs = x if x.dtype == torch.float32 else x.to(torch.float32)
        stride = xs.stride(0) if xs.numel() > 1 else 1
        return stft_dev(xs, n_fft, step, window, zeropad, _mode, x_stride=max(stride, 1), n=xs.shape[0], dev=dev)
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError('x must be 1D')
    window = dsp.get_window(window_name, n_fft).astype(np.float32)
    return hip_rfft2(n_fft, step, window, x, zeropad, _mode)


def istft_dev(spec_t, hop_length, window_t, length=None, dev=None):
    """spec_t: complex64 tensor with logical shape (bins, frames), frame-major memory preferred."""
    dev = _dev.device_index(dev if dev is not None else spec_t.device)
    L = _lib.lib()
    bins, total_frames = spec_t.shape
    n_fft = 2 * (bins - 1)
    fm = spec_t.T.contiguous()                      # [frames][bins]; no copy when it came from stft_dev
    if fm.dtype != torch.complex64:
        fm = fm.to(torch.complex64)
    if length:
        n_frames = min(total_frames, int(np.ceil((length + n_fft) / hop_length)))
    else:
        n_frames = total_frames
    scratch = int(L.par_istft_scratch_floats(n_frames, n_fft, hop_length))        # 0: frames are overlap-added in LDS
    frames = _dev.empty(scratch, torch.float32, dev) if scratch else None
    if length is None:
        y_len = hop_length * (n_frames - 1)         # y[n_fft//2 : -(n_fft//2)]
    else:
        y_len = int(length)
    y = _dev.empty(max(y_len, 0), torch.float32, dev)
    if y.numel() == 0:                              # a single frame without an explicit length trims to nothing
        return y
    _lib.check(L.par_istft_f32(dev, _dev.ptr(fm), n_frames, n_fft, hop_length, _dev.ptr(window_t), _dev.ptr(frames) if scratch else None,
                               _dev.ptr(y), y.numel(), n_fft // 2, _dev.stream_ptr(dev)))
    return y


def istft(stft_matrix, hop_length=None, win_length=None, window_name='blackmanharris', center=True, dtype=None,
          length=None):
    """Inverse STFT (least squares, window-sumsquare normalised), reference util/fourier.py:314-437.

    Differences, both documented in DESIGN.md: the argument is NOT mutated (the reference scales it
    by sqrt(n_fft) in place) and the result is float32 unless `dtype` says otherwise.
    Only the shipped call pattern is supported: center=True, win_length in (None, n_fft)."""
    n_fft = 2 * (stft_matrix.shape[0] - 1)
    if win_length is None:
        win_length = n_fft
    if not center or win_length != n_fft:
        raise NotImplementedError("HIP istft supports center=True and win_length == n_fft (the shipped call sites)")
    if hop_length is None:
        hop_length = int(win_length // 4)
    if isinstance(stft_matrix, torch.Tensor):
        dev = _dev.device_index(stft_matrix.device)
        return istft_dev(stft_matrix, hop_length, window_dev(window_name, win_length, dev), length, dev)
    window = dsp.get_window(window_name, win_length, fftbins=True).astype(np.float32)
    dev = _dev.device_index(None)
    S = np.asarray(stft_matrix)
    spec_t = _dev.to_dev(S.T, torch.complex64, dev).T
    y = istft_dev(spec_t, hop_length, _dev.to_dev(window, torch.float32, dev), length, dev)
    y = _dev.to_host(y)
    if dtype is not None:
        y = y.astype(dtype)
    return y


def fix_length(data, size, axis=-1, **kwargs):
    """Trim or trailing-pad `data` to exactly `size` along `axis` (np.pad keywords, zero padding by default);
    a view when trimming, the array itself when the length already fits -- util/fourier.py:440-478."""
    have = data.shape[axis]
    if have == size:
        return data
    ax = axis % data.ndim
    if have > size:
        return data[(slice(None),) * ax + (slice(0, size),)]
    widths = [(0, size - have) if k == ax else (0, 0) for k in range(data.ndim)]
    return np.pad(data, widths, **{"mode": "constant", **kwargs})


def fft_freqs(n_fft, fs):
    """Return frequencies for DFT"""
    return np.arange(0, (n_fft // 2 + 1)) / float(n_fft) * float(fs)
