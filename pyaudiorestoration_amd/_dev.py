"""Device plumbing: PyTorch-ROCm is used ONLY to own HBM buffers and name streams."""
import ctypes
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


def device_index(device=None):
    """Resolve a device argument to an ordinal.  Raises when no GPU is visible: the
    product path never falls back to the CPU."""
    if not torch.cuda.is_available():
        raise RuntimeError("pyaudiorestoration_amd: no ROCm GPU visible (torch.cuda.is_available() is False); "
                           "the HIP path has no CPU fallback")
    if device is None:
        return torch.cuda.current_device()
    if isinstance(device, int):
        return device
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


def stream_ptr(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


# ---- large pageable host arrays <-> HBM through a ring of pinned chunks ------------------------------------------------
# The reference hands numpy arrays over (operator slots: util/fourier.py:67-70, util/resampling.py:21-46).  A plain copy from
# pageable memory runs at ~12 GB/s of the bus's 57: the runtime stages it through its own pinned buffer on ONE thread.  Here the
# staging is ours: four threads fill (or drain) 16 MB pinned chunks -- numpy's copy releases the GIL -- while earlier chunks are on
# the bus.  Arrays below 64 MB take the plain copy.
_STAGE_MIN = 64 << 20
_STAGE_CHUNK = 16 << 20
_STAGE_RING = 6
_stage = {}
_stage_lock = threading.Lock()


def _stage_state(dev, kind):
    st = _stage.get((dev, kind))
    if st is None:
        ring = [torch.empty(_STAGE_CHUNK, dtype=torch.uint8).pin_memory() for _ in range(_STAGE_RING)]
        st = _stage[(dev, kind)] = {"ring": ring, "np": [r.numpy() for r in ring], "busy": threading.Lock()}
    if "pool" not in _stage:
        _stage["pool"] = ThreadPoolExecutor(max_workers=4, thread_name_prefix="par_stage")
    return st


def _h2d_staged(src, dst, dev):
    """src: contiguous numpy array, dst: contiguous device tensor of the same byte size -> False when the ring is in use."""
    with _stage_lock:
        st = _stage_state(dev, "up")
    if not st["busy"].acquire(blocking=False):
        return False
    try:
        src_u8, dst_u8 = src.reshape(-1).view(np.uint8), dst.view(-1).view(torch.uint8)
        n, C, R, pool = src_u8.size, _STAGE_CHUNK, _STAGE_RING, _stage["pool"]
        stream = torch.cuda.current_stream(dev)
        events = [None] * R

        def fill(k):
            b, lo = k % R, k * C
            hi = min(n, lo + C)
            if events[b] is not None:
                events[b].synchronize()                # the bus is done with what this chunk buffer held (chunk k - R)
            np.copyto(st["np"][b][:hi - lo], src_u8[lo:hi])
            return lo, hi, b
        nch = -(-n // C)
        futs, submitted = [None] * nch, 0
        for k in range(nch):
            while submitted < nch and submitted < k + R:   # fills run at most R chunks ahead of the copies issued: chunk j - R has its event
                futs[submitted] = pool.submit(fill, submitted)
                submitted += 1
            lo, hi, b = futs[k].result()
            dst_u8[lo:hi].copy_(st["ring"][b][:hi - lo], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
            events[b] = ev
        for ev in events:
            if ev is not None:
                ev.synchronize()                           # the ring may be refilled by the next call
        return True
    finally:
        st["busy"].release()


def _d2h_staged(src, dst, dev):
    """src: contiguous device tensor, dst: contiguous numpy array of the same byte size -> False when the ring is in use."""
    with _stage_lock:
        st = _stage_state(dev, "down")
    if not st["busy"].acquire(blocking=False):
        return False
    try:
        src_u8, dst_u8 = src.view(-1).view(torch.uint8), dst.reshape(-1).view(np.uint8)
        n, C, R, pool = dst_u8.size, _STAGE_CHUNK, _STAGE_RING, _stage["pool"]
        stream = torch.cuda.current_stream(dev)

        def drain(lo, hi, b, ev):
            ev.synchronize()
            np.copyto(dst_u8[lo:hi], st["np"][b][:hi - lo])
        nch = -(-n // C)
        futs = [None] * nch
        for k in range(nch):
            b, lo = k % R, k * C
            hi = min(n, lo + C)
            if k >= R:
                futs[k - R].result()                       # chunk k - R has left this buffer
            st["ring"][b][:hi - lo].copy_(src_u8[lo:hi], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
            futs[k] = pool.submit(drain, lo, hi, b, ev)
        for f in futs[max(0, nch - R):]:
            f.result()
        return True
    finally:
        st["busy"].release()


def host_copy(dst, src):
    """dst[...] = src for two contiguous host arrays (numpy or CPU tensors) of one size, split over the staging threads: a pinned
    slot is filled at the host's memory bandwidth rather than one core's (460 MB: 46 -> ~12 ms)."""
    d = (dst.numpy() if isinstance(dst, torch.Tensor) else dst).reshape(-1).view(np.uint8)
    s = (src.numpy() if isinstance(src, torch.Tensor) else src).reshape(-1).view(np.uint8)
    if d.size != s.size:
        raise ValueError("host_copy: sizes differ")
    if d.size < (8 << 20):
        np.copyto(d, s)
        return
    with _stage_lock:
        if "pool" not in _stage:
            _stage["pool"] = ThreadPoolExecutor(max_workers=4, thread_name_prefix="par_stage")
        pool = _stage["pool"]
    step = -(-d.size // 8) + 63 & ~63
    list(pool.map(lambda lo: np.copyto(d[lo:lo + step], s[lo:lo + step]), range(0, d.size, step)))


def _pool():
    with _stage_lock:
        if "pool" not in _stage:
            _stage["pool"] = ThreadPoolExecutor(max_workers=4, thread_name_prefix="par_stage")
        return _stage["pool"]


def host_assign(dst, src):
    """dst[...] = src for numpy arrays of one shape, any strides and dtypes -- rows split over the staging threads when the arrays
    are large.  The reference hands over and takes back COLUMN VIEWS of interleaved files (util/resampling.py:222-227:
    output[:, k], signal[:, ch]): one core gathers / scatters such a view at ~2 GB/s, 0.23 s per 10-min channel each way, five
    times what the bus and the kernels take together."""
    if dst.shape != src.shape:
        raise ValueError("host_assign: shapes differ")
    n = dst.shape[0] if dst.ndim else 0
    if dst.nbytes < (32 << 20) or n < 16:
        np.copyto(dst, src, casting="unsafe")
        return
    step = -(-n // 8)
    list(_pool().map(lambda lo: np.copyto(dst[lo:lo + step], src[lo:lo + step], casting="unsafe"), range(0, n, step)))


def contiguous(a, dtype=None):
    """np.ascontiguousarray(a, dtype) with the copy (if one is needed) on the staging threads."""
    dtype = np.dtype(a.dtype if dtype is None else dtype)
    if (a.flags.c_contiguous and a.dtype == dtype) or a.nbytes < (32 << 20):
        return np.ascontiguousarray(a, dtype=dtype)
    out = np.empty(a.shape, dtype=dtype)
    host_assign(out, a)
    return out


def to_host(t, out=None):
    """Device tensor -> numpy array (`out`: a contiguous array of the same shape and dtype to fill).  Large tensors leave through
    the pinned ring."""
    if out is None and t.ndim == 2 and not t.is_contiguous() and t.T.is_contiguous():
        return to_host(t.T).T           # a transposed view (stft_dev's frame-major spectrogram) keeps its memory order, like .cpu()
    t = t.contiguous()
    np_dtype = torch.empty(0, dtype=t.dtype).numpy().dtype
    if out is None:
        out = np.empty(tuple(t.shape), dtype=np_dtype)
    nbytes = t.numel() * t.element_size()
    if not (t.is_cuda and nbytes >= _STAGE_MIN and out.flags.c_contiguous and out.dtype == np_dtype and out.nbytes == nbytes and
            _d2h_staged(t, out, t.device.index)):
        out[...] = t.cpu().numpy()
    return out


def to_dev(a, dtype, dev):
    """numpy (any stride) / torch -> contiguous device tensor of `dtype`.  Large numpy arrays go up through the pinned ring."""
    if isinstance(a, torch.Tensor):
        return a.to(device=f"cuda:{dev}", dtype=dtype).contiguous()
    np_dtype = {torch.float32: np.float32, torch.float64: np.float64, torch.complex64: np.complex64,
                torch.int32: np.int32, torch.int64: np.int64}[dtype]
    a = np.asarray(a)
    if a.dtype in (np.float32, np.int32, np.int16, np.int8, np.uint8) and a.dtype.itemsize < np.dtype(np_dtype).itemsize:
        # an exact widening (float32 -> float64, int16 -> float32 ...): upload the narrow form and widen in HBM -- half the
        # bytes over PCIe and no host-side conversion pass (ZeroCrossingTracker on 2 min at 192 kHz: 36-41 -> 19 ms)
        return _upload(contiguous(a), dev).to(dtype)
    return _upload(contiguous(a, np_dtype), dev)


def _upload(a, dev):
    if a.nbytes >= _STAGE_MIN:
        out = torch.empty(a.shape, dtype=torch.from_numpy(a[:0].reshape(-1)).dtype, device=f"cuda:{dev}")
        if _h2d_staged(a, out, dev):
            return out
    return torch.from_numpy(a).to(f"cuda:{dev}")


def empty(shape, dtype, dev):
    return torch.empty(shape, dtype=dtype, device=f"cuda:{dev}")


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())
