"""ctypes loader for oracle/libpar_oracle.so (the plain-C CPU restatement).

TEST INFRASTRUCTURE ONLY -- see par_oracle.c.  Used by tests/ and bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpar_oracle.so")
_lib = None

i64, dbl, vp, ci = ctypes.c_int64, ctypes.c_double, ctypes.c_void_p, ctypes.c_int


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libpar_oracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.oracle_speed_to_pos.restype = ci
        L.oracle_speed_to_pos.argtypes = [vp, vp, i64, i64, vp, i64, ctypes.POINTER(i64), ctypes.POINTER(ci)]
        L.oracle_speed_to_pos_windows.restype = ci
        L.oracle_speed_to_pos_windows.argtypes = [vp, vp, i64, i64, vp, ci, i64, vp, ctypes.POINTER(i64), ctypes.POINTER(ci), ci]
        L.oracle_end_guess.restype = i64
        L.oracle_end_guess.argtypes = [vp, vp, i64]
        L.oracle_sinc.restype = ci
        L.oracle_sinc.argtypes = [vp, i64, vp, i64, i64, ci, vp, i64]
        L.oracle_sinc_mt.restype = ci
        L.oracle_sinc_mt.argtypes = [vp, i64, vp, i64, i64, ci, vp, i64, ci]
        L.oracle_stft.restype = ci
        L.oracle_stft.argtypes = [vp, i64, i64, ci, ci, ci, vp, vp, ci]
        L.oracle_stft_mt.restype = ci
        L.oracle_stft_mt.argtypes = [vp, i64, i64, ci, ci, ci, vp, vp, ci, ci]
        L.oracle_synth_signal.restype = None
        L.oracle_synth_signal.argtypes = [vp, i64, i64, dbl, ctypes.c_uint64]
        L.oracle_synth_curve.restype = None
        L.oracle_synth_curve.argtypes = [vp, vp, i64, dbl, dbl, dbl, dbl, dbl]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(vp)


def speed_to_pos(sampletimes, speeds, n_in):
    st = np.ascontiguousarray(sampletimes, dtype=np.float64)
    sp = np.ascontiguousarray(speeds, dtype=np.float64)
    L = lib()
    cap = int(L.oracle_end_guess(_p(st), _p(sp), len(st)))
    out = np.empty(max(cap, 1), dtype=np.float64)
    n = i64(0)
    tr = ci(0)
    rc = L.oracle_speed_to_pos(_p(st), _p(sp), len(st), int(n_in), _p(out), cap, ctypes.byref(n), ctypes.byref(tr))
    if rc != 0:
        raise ValueError(f"oracle_speed_to_pos status {rc}")
    return out[:n.value].copy(), bool(tr.value)


def speed_to_pos_windows(sampletimes, speeds, n_in, starts, width, threads=1):
    """(positions [len(starts)][width] of the outputs starts[w] .. starts[w] + width - 1, len_out, trimmed): the same values as
    speed_to_pos(...)[0][s:s + width] (NaN at and beyond len_out) without the whole array; see par_oracle.c"""
    st = np.ascontiguousarray(sampletimes, dtype=np.float64)
    sp = np.ascontiguousarray(speeds, dtype=np.float64)
    ss = np.ascontiguousarray(starts, dtype=np.int64)
    out = np.empty((len(ss), int(width)), dtype=np.float64)
    n = i64(0)
    tr = ci(0)
    rc = lib().oracle_speed_to_pos_windows(_p(st), _p(sp), len(st), int(n_in), _p(ss), len(ss), int(width), _p(out),
                                           ctypes.byref(n), ctypes.byref(tr), int(threads))
    if rc != 0:
        raise ValueError(f"oracle_speed_to_pos_windows status {rc}")
    return out, n.value, bool(tr.value)


def sinc(pos, sig, NT, threads=1):
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    sig = np.ascontiguousarray(sig, dtype=np.float32)
    out = np.empty(len(pos), dtype=np.float32)
    L = lib()
    if threads == 1:
        rc = L.oracle_sinc(_p(pos), len(pos), _p(sig), 1, len(sig), int(NT), _p(out), 1)
    else:
        rc = L.oracle_sinc_mt(_p(pos), len(pos), _p(sig), 1, len(sig), int(NT), _p(out), 1, int(threads))
    if rc != 0:
        raise ValueError(f"oracle_sinc status {rc}")
    return out


def stft(x, n_fft, hop, window, zeropad=1, mode=0, threads=1):
    x = np.ascontiguousarray(x, dtype=np.float32)
    window = np.ascontiguousarray(window, dtype=np.float32)
    bins = n_fft * zeropad // 2 + 1
    frames = (len(x) + 2 * (n_fft // 2) - n_fft) // hop + 1
    out = np.empty((frames, bins * (2 if mode == 0 else 1)), dtype=np.float32)
    if threads > 1:
        rc = lib().oracle_stft_mt(_p(x), len(x), 1, n_fft, hop, zeropad, _p(window), _p(out), mode, int(threads))
    else:
        rc = lib().oracle_stft(_p(x), len(x), 1, n_fft, hop, zeropad, _p(window), _p(out), mode)
    if rc != 0:
        raise ValueError(f"oracle_stft status {rc}")
    if mode == 0:
        return out.view(np.complex64).T
    return out.T


def synth_signal(start, count, sr, seed=0x5EED):
    out = np.empty(count, dtype=np.float32)
    lib().oracle_synth_signal(_p(out), int(start), int(count), float(sr), ctypes.c_uint64(seed))
    return out


def synth_curve(m, dur, sr, depth=0.01, rate=0.55, phase=0.7):
    st = np.empty(m, dtype=np.float64)
    sp = np.empty(m, dtype=np.float64)
    lib().oracle_synth_curve(_p(st), _p(sp), int(m), float(dur), float(sr), depth, rate, phase)
    return st, sp
