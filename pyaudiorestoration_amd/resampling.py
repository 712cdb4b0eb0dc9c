"""Mirror of reference util/resampling.py -- same names, arguments and behaviour, HIP inside.

sinc_wrapper :21-27, sinc_wrapper_mt :30-46, sinc_core :51-90, speed_to_pos :93-137,
find_cutoff :14-18, run :162-240.  The `*_dev` functions keep everything in HBM (torch tensors
only own the buffers); the numpy-signature functions wrap them with H2D/D2H copies.
"""
import ctypes
import logging
import os
import threading
from time import time

import numpy as np
import torch

from . import _dev, _lib
from .timing import log_duration


def find_cutoff(array, cutoff):
    """First index with array[idx] >= cutoff, as a 1-tuple like np.ndenumerate yields; else None."""
    hit = np.nonzero(np.asarray(array) >= cutoff)[0]
    return (int(hit[0]),) if len(hit) else None


# ----------------------------------------------------------------------------- positions

class SpeedPlan:
    """Device-resident result of the planning stage of speed_to_pos (segment lengths, offsets, trim)."""

    def __init__(self, speeds_t, m, work, len_out, trimmed, path, dev, aux=None, fused_ok=False, max_out=0, lazy=False):
        self.speeds_t, self.m, self.work, self.len_out, self.trimmed, self.path, self.dev = \
            speeds_t, m, work, len_out, trimmed, path, dev
        # fused_ok: the fused K_sinc can run from this plan.  lazy: it was made without the per-sample cumsum (closed-form
        # segment sums, exact ones only where a rounding decides: csrc/pos_plan.h) -- same positions and window centres, but
        # no checkpoints, so par_speed_to_pos_fill_fused cannot fill from it (ask for eager=True then)
        self.aux, self.fused_ok, self.max_out, self.lazy = aux, fused_ok, max_out, lazy


def _max_out_for_bytes(L, nbytes, m):
    """Largest output bound whose fused aux layout fits nbytes (par_fused_aux_bytes is monotone in max_out)."""
    lo, hi = 0, max(int(nbytes) // 2, 1)               # > 2 bytes of aux per output: hi never fits... unless tiny
    if int(L.par_fused_aux_bytes(hi, m)) <= nbytes:
        return hi
    while hi - lo > 1024:
        mid = (lo + hi) // 2
        if int(L.par_fused_aux_bytes(mid, m)) <= nbytes:
            lo = mid
        else:
            hi = mid
    return lo


def fused_max_out(sampletimes_t, speeds_t):
    """Output bound used to size the fused plan's checkpoint buffer: the reference's own end_guess
    (util/resampling.py:108, int(mean(speeds) * span * 1.01)) plus the longest segment.  Buffer sizing only,
    so torch reductions are fine here."""
    span = float(sampletimes_t[-1] - sampletimes_t[0])
    longest = float((sampletimes_t[1:] - sampletimes_t[:-1]).max()) * float(speeds_t.max())
    return int(float(speeds_t.mean()) * span * 1.01) + int(longest) + 1024


def speed_plan_dev(sampletimes_t, speeds_t, num_imput_samples, dev=None, force_host_chain=False, fused=False,
                   max_out=None, work=None, aux=None, stream=None, eager=False):
    """Planning stage.  fused=True also stores per-segment cumsum checkpoints (every 8th step) so that
    varispeed_resample_dev can regenerate positions inside K_sinc instead of reading a position array.
    work / aux: caller-owned uint8 device buffers to (re)use; stream: torch stream to plan on (default: current).
    force_host_chain: True/1 = serial host evaluation of the two chains; 2 (tests) = that plus an injected checkpoint
    verification failure, after which fused_ok must be False.
    eager: always compute the per-sample cumsum and its checkpoints (default: dense gentle curves get a lazy plan)."""
    dev = _dev.device_index(dev if dev is not None else sampletimes_t.device)
    L = _lib.lib()
    m = sampletimes_t.numel()
    if speeds_t.numel() != m:
        raise ValueError("sampletimes and speeds must have the same length")
    nbytes = int(L.par_speed_plan_bytes(m))
    if work is None or work.numel() < nbytes:
        work = _dev.empty(nbytes, torch.uint8, dev)
    s_ptr = _dev.stream_ptr(dev) if stream is None else ctypes.c_void_p(stream.cuda_stream)
    len_out = ctypes.c_int64(0)
    trimmed = ctypes.c_int(0)
    path = ctypes.c_int(0)
    if fused:
        def plan_with(max_out_, aux_):
            ok = ctypes.c_int(0)
            _lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(sampletimes_t), _dev.ptr(speeds_t), m,
                                                     int(num_imput_samples), _dev.ptr(work), work.numel(), _dev.ptr(aux_),
                                                     aux_.numel(), max_out_, ctypes.byref(len_out), ctypes.byref(trimmed),
                                                     int(force_host_chain) | (8 if eager else 0), ctypes.byref(path),
                                                     ctypes.byref(ok), s_ptr))
            return SpeedPlan(speeds_t, m, work, len_out.value, bool(trimmed.value), path.value, dev, aux_, bool(ok.value),
                             max_out_, lazy=ok.value == 2)

        if max_out is None and aux is not None:
            # A batch hands the previous item's buffer back: size the plan to what that buffer holds instead of running
            # three device reductions with host read-backs for the bound (under a concurrent K_sinc each of them waits
            # ~0.5 ms for a free CU: they, not the kernels, set the pace of a batch of 10-minute files).  A buffer that
            # turns out too small makes the plan refuse the fused form: only then the bound is computed and the plan redone.
            cap = _max_out_for_bytes(L, aux.numel(), m)
            if cap > 0:
                plan = plan_with(cap, aux)
                # k_tile_seg's spare tiles and the checkpoint slack let a curve a little LONGER than cap still come
                # back with valid checkpoints; K_sinc requires len_out <= max_out, so such a plan is redone below
                if plan.fused_ok and plan.len_out <= cap:
                    return plan
        if max_out is None:
            max_out = fused_max_out(sampletimes_t, speeds_t)
        aux_bytes = int(L.par_fused_aux_bytes(max_out, m))
        if aux is None or aux.numel() < aux_bytes:
            aux = _dev.empty(aux_bytes, torch.uint8, dev)
        return plan_with(max_out, aux)
    _lib.check(L.par_speed_to_pos_plan_ex(dev, _dev.ptr(sampletimes_t), _dev.ptr(speeds_t), m, int(num_imput_samples),
                                          _dev.ptr(work), work.numel(), ctypes.byref(len_out), ctypes.byref(trimmed),
                                          1 if force_host_chain else 0, ctypes.byref(path), s_ptr))
    return SpeedPlan(speeds_t, m, work, len_out.value, bool(trimmed.value), path.value, dev)


def speed_to_pos_dev(sampletimes_t, speeds_t, num_imput_samples, dev=None, force_host_chain=False, info=None):
    """Device speed curve (float64 tensors) -> float64 position tensor (the written prefix).
    force_host_chain runs the serial host evaluation of the two order-dependent chains (the exact
    fallback of the device scans); `info`, if a dict, receives {"path": 0 device | 1 host, "trimmed"}.
    Curves with few points (segments of thousands of samples and more) go through a fused plan: its cumsum
    checkpoints let the fill run in parallel over 8-sample blocks instead of one lane per segment."""
    m = sampletimes_t.numel()
    sparse = m >= 2 and int(num_imput_samples) // max(m - 1, 1) > 16384
    plan = speed_plan_dev(sampletimes_t, speeds_t, num_imput_samples, dev, force_host_chain, fused=sparse, eager=True)
    if info is not None:
        info.update(path=plan.path, trimmed=plan.trimmed)
    pos = _dev.empty(plan.len_out, torch.float64, plan.dev)
    L = _lib.lib()
    if sparse and plan.fused_ok and not plan.lazy:
        _lib.check(L.par_speed_to_pos_fill_fused(plan.dev, _dev.ptr(speeds_t), plan.m, _dev.ptr(plan.work), _dev.ptr(plan.aux),
                                                 plan.max_out, _dev.ptr(pos), plan.len_out, _dev.stream_ptr(plan.dev)))
    else:
        _lib.check(L.par_speed_to_pos_fill(plan.dev, _dev.ptr(speeds_t), plan.m, _dev.ptr(plan.work), _dev.ptr(pos),
                                           plan.len_out, _dev.stream_ptr(plan.dev)))
    return pos


def varispeed_fused_dev(plan, sig_t, NT, out_t=None, sig_stride=1, len_in=None, out_stride=1):
    """Sinc interpolation of one channel straight from a fused SpeedPlan: K_sinc places every output itself from the plan's
    block records (no position array in HBM).  Same output as the two-step path within the contract's tolerance, every window
    centre the reference's.  NT = 32 on unit strides, or one channel of a two-channel interleaved file (sig_stride=2: the
    reference's use_channels views), takes the streaming kernel, everything else the block kernel (include/par_hip.h)."""
    if not plan.fused_ok:
        raise ValueError("plan cannot feed the fused resampler: build it with speed_plan_dev(..., fused=True)")
    dev = plan.dev
    L = _lib.lib()
    if len_in is None:
        len_in = sig_t.numel() // sig_stride
    if out_t is None:
        out_t = _dev.empty(plan.len_out * out_stride, torch.float32, dev)
    _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(plan.speeds_t), plan.m, _dev.ptr(plan.work), _dev.ptr(plan.aux),
                                         plan.max_out, plan.len_out, _dev.ptr(sig_t), sig_stride, len_in, int(NT),
                                         _dev.ptr(out_t), out_stride, _dev.stream_ptr(dev)))
    return out_t


def varispeed_fused_stereo_dev(plan, sig0_t, sig1_t, NT, out0_t, out1_t, sig_stride=1, len_in=None, out_stride=1):
    """Two channels of one file in ONE fused K_sinc launch (positions, prologue and tap weights shared).  sig0/sig1 and
    out0/out1 are channel views with common strides, e.g. the two columns of an interleaved (n, 2) tensor.  Equal to
    two varispeed_fused_dev calls up to float32 rounding (different lane <-> output map)."""
    if not plan.fused_ok:
        raise ValueError("plan has no valid checkpoints: build it with speed_plan_dev(..., fused=True)")
    if len_in is None:
        len_in = sig0_t.numel() // sig_stride
    _lib.check(_lib.lib().par_varispeed_fused_stereo_f32(
        plan.dev, _dev.ptr(plan.speeds_t), plan.m, _dev.ptr(plan.work), _dev.ptr(plan.aux), plan.max_out, plan.len_out,
        _dev.ptr(sig0_t), _dev.ptr(sig1_t), sig_stride, len_in, int(NT), _dev.ptr(out0_t), _dev.ptr(out1_t), out_stride,
        _dev.stream_ptr(plan.dev)))
    return out0_t, out1_t


def _resample_item(plan, item, NT, dev):
    """K_sinc launch(es) of one planned work item on the current stream: (.., sig_t[, sig_stride, len_in]) with a 1-D
    sig_t, or an interleaved (n, ch) sig_t whose channel pairs share one stereo launch (an odd last channel goes
    alone).  A plan without valid checkpoints takes the position-array path."""
    sig_t = item[2]
    if sig_t.ndim == 2:
        n_in, ch = sig_t.shape
        out_t = _dev.empty((plan.len_out, ch), torch.float32, dev)
        flat_in, flat_out = sig_t.reshape(-1), out_t.reshape(-1)
        layout = dict(sig_stride=ch, len_in=n_in, out_stride=ch)
        if not plan.fused_ok:
            pos_t = _dev.empty(plan.len_out, torch.float64, dev)
            _lib.check(_lib.lib().par_speed_to_pos_fill(dev, _dev.ptr(plan.speeds_t), plan.m, _dev.ptr(plan.work),
                                                        _dev.ptr(pos_t), plan.len_out, _dev.stream_ptr(dev)))
        c = 0
        while c < ch:
            if plan.fused_ok and c + 1 < ch:
                varispeed_fused_stereo_dev(plan, flat_in[c:], flat_in[c + 1:], NT, flat_out[c:], flat_out[c + 1:], **layout)
                c += 2
            elif plan.fused_ok:
                varispeed_fused_dev(plan, flat_in[c:], NT, flat_out[c:], **layout)
                c += 1
            else:
                sinc_resample_dev(pos_t, flat_in[c:], NT, flat_out[c:], dev=dev, **layout)
                c += 1
        return out_t
    stride = item[3] if len(item) > 3 else 1
    len_in = item[4] if len(item) > 4 else sig_t.numel() // stride
    if plan.fused_ok:
        return varispeed_fused_dev(plan, sig_t, NT, sig_stride=stride, len_in=len_in)
    return varispeed_resample_dev(plan, sig_t, NT, sig_stride=stride, len_in=len_in)[0]


def _group_class(item):
    """Items whose K_sinc launches par_varispeed_fused_batch_f32 can merge: 'mono' (1-D, unit stride), 'stereo' (an interleaved
    (n, 2) file); None: launched on its own."""
    sig_t = item[2]
    if sig_t.ndim == 2:
        return "stereo" if sig_t.shape[1] == 2 and sig_t.is_contiguous() else None
    return "mono" if len(item) <= 3 or item[3] == 1 else None


def _group_size(item):
    """How many files of this class and size go into one launch.  Measured (r06, tools/exp/group_speed.py; G samples/s):
    K_sinc alone on ready plans gains from merging -- 10-s mono files 22 -> 85, 60-s 80 -> 152, 10-min mono 167 -> 183, 10-min
    stereo 205 -> 208 -- because a file's K_sinc is two or three kernels with a tail of idle wave slots each.  Through the batch
    driver (plans included) what is left of that: 10-min mono 132 -> 141 in groups of four, 60-s mono 50 -> 58 (the host side
    of a plan, ~0.2 ms per file, bounds short files); stereo files LOSE (archive 182 -> 178 in pairs, 170 in fours: the plans
    of a group start together and then compete with one long launch instead of slipping into the gaps between short ones); so
    do 60-min files.  Hence: stereo and long mono files one by one, other mono files four at a time."""
    sig_t = item[2]
    if sig_t.ndim == 2 or sig_t.numel() >= 400_000_000:
        return 1
    return 4


def _resample_group(plans, group, NT, dev):
    """ONE merged K_sinc launch (par_varispeed_fused_batch_f32) for the planned items of a group -> their output tensors."""
    L = _lib.lib()
    arr = (_lib.FusedItem * len(group))()
    outs = []
    for k, (plan, item) in enumerate(zip(plans, group)):
        sig_t = item[2]
        f = arr[k]
        f.speeds, f.m, f.work, f.aux = _dev.ptr(plan.speeds_t).value, plan.m, _dev.ptr(plan.work).value, _dev.ptr(plan.aux).value
        f.max_out, f.len_out = plan.max_out, plan.len_out
        if sig_t.ndim == 2:
            out_t = _dev.empty((plan.len_out, 2), torch.float32, dev)
            base, obase = _dev.ptr(sig_t).value, _dev.ptr(out_t).value
            f.sig0, f.sig1, f.sig_stride, f.len_in = base, base + 4, 2, sig_t.shape[0]
            f.out0, f.out1, f.out_stride = obase, obase + 4, 2
        else:
            out_t = _dev.empty(plan.len_out, torch.float32, dev)
            f.sig0, f.sig1, f.sig_stride, f.len_in = _dev.ptr(sig_t).value, None, 1, (item[4] if len(item) > 4 else sig_t.numel())
            f.out0, f.out1, f.out_stride = _dev.ptr(out_t).value, None, 1
        outs.append(out_t)
    _lib.check(L.par_varispeed_fused_batch_f32(dev, len(group), arr, int(NT), _dev.stream_ptr(dev)))
    return outs


# plan / aux buffers, slot events and planner streams of varispeed_batch_dev, kept per device between calls: a call that had to
# allocate them afresh (hipMalloc of 6-32 buffers of 0.15-0.9 GB, on new streams whose allocator pools are empty) paid for it
# inside its first files -- 48-file runs measured 4 % below 512-file ones (r06)
_plan_rings = {}
_plan_rings_lock = threading.Lock()


def _borrow_plan_ring(dev, n_slots, P):
    """The device's cached ring, grown to n_slots slots and P planner streams; a private one when the cached ring is in use
    (two batch drivers at once on one device).  The slot events carry over: a plan of the next call still waits for the
    K_sinc of the previous call that read its slot."""
    with _plan_rings_lock:
        ring = _plan_rings.get(dev)
        if ring is None or ring["busy"]:
            ring = {"busy": False, "sides": [], "work": [], "aux": [], "free": []}
            if dev not in _plan_rings:
                _plan_rings[dev] = ring
        ring["busy"] = True
    while len(ring["sides"]) < P:
        ring["sides"].append(torch.cuda.Stream(device=dev))
    for key in ("work", "aux", "free"):
        ring[key].extend([None] * (n_slots - len(ring[key])))
    return ring


def release_plan_rings():
    """Free the plan buffers varispeed_batch_dev keeps between calls (2 x lookahead slots of ~1.3 B per output sample each)."""
    with _plan_rings_lock:
        for dev in [d for d, r in _plan_rings.items() if not r["busy"]]:
            del _plan_rings[dev]


def varispeed_batch_dev(items, NT, dev=None, planners=None, group=None):
    """Software-pipelined fused resampling of a batch of device-resident work items on one GPU (the per-GPU
    inner loop of a file batch).  K_sinc launches follow one another on the current stream; the plans
    of the next `planners` items -- each ~20 small latency-bound kernels and a header read-back, 0.6 ms on an idle
    GPU -- are made by as many planner threads, each on its own side stream.  Beside the block kernel a plan runs
    under the K_sinc in front of it; the streaming kernel (mono NT = 32) leaves it no room, so plans only advance
    in the gaps around a K_sinc -- where `planners` of them then advance TOGETHER (they are latency-, not
    throughput-bound).  Measured on 60-min mono files (r05, ms per file): 1 planner 5.0, 2: 4.66, 3: 4.57, 4: 4.63, 8: 4.63
    -- default 3 (PAR_PLANNERS).
    Plan buffers form a ring of 2 x planners slots; an event keeps a slot from being re-planned before the K_sinc
    that reads it has finished.  planners=1 is the double-buffered pipeline of r02-r04.

    PREFETCH CONTRACT: `planners` items (default 3) are taken from the iterable AHEAD of the one being resampled, and their
    tensors are read until that item's output has been yielded -- a producer that recycles its input tensors needs a ring of
    at least planners + 1 of them (planners=1 is the one-ahead contract of r02-r04).  Memory: 2 x planners plan / aux buffer
    pairs, each sized for the eager plan's checkpoints (~1.3 B per output sample: 0.9 GB for a 60-min file), stay allocated
    BETWEEN calls as well (release_plan_rings() frees them).

    GROUPS (r06): consecutive items of one class -- mono on unit strides, or interleaved stereo files -- are resampled by ONE merged
    K_sinc launch per group (par_varispeed_fused_batch_f32; default: mono files below 400 M samples four at a time, everything
    else one by one -- _group_size has the measurements; `group` = 1..8 overrides): a file's K_sinc is two or three kernels,
    each ending in a tail of idle wave slots.  Outputs are bit-identical to the ungrouped launches; a group's
    items are yielded together once its launch has been queued.  The prefetch contract grows to max(planners + group - 1, 2 x group)
    items taken ahead (the next group is planned under the launch of this one): a recycling producer needs that many + group tensors.

    items: iterable of (sampletimes_t, speeds_t, sig_t) or (sampletimes_t, speeds_t, sig_t, sig_stride, len_in)
    with float64 / float32 device tensors; a 2-D sig_t is an interleaved (n, ch) file whose channels share the plan
    (channel pairs go through the stereo launch) and yields an (len_out, ch) output.  Up to `planners` items are
    taken from the iterable ahead of the one being resampled (their tensors stay resident meanwhile).  Yields
    (index, out_t, plan) in order; out_t is ready on the current stream (synchronise or keep using that stream).
    An item whose plan cannot feed the fused resampler is resampled through the position-array path."""
    import collections
    from concurrent.futures import ThreadPoolExecutor
    dev = _dev.device_index(dev)
    P = planners if planners is not None else os.environ.get("PAR_PLANNERS", "3")
    try:
        P = int(P)
    except (TypeError, ValueError):
        raise ValueError(f"planners / PAR_PLANNERS must be an integer 1..8, got {P!r}")
    if not 1 <= P <= 8:
        raise ValueError(f"planners / PAR_PLANNERS must be an integer 1..8, got {P}")
    if group is None and os.environ.get("PAR_GROUP"):        # (A/B sessions: 1 = the ungrouped launches of r05)
        group = os.environ["PAR_GROUP"]
    try:
        group = None if group is None else int(group)
    except (TypeError, ValueError):
        raise ValueError(f"group / PAR_GROUP must be an integer 1..8, got {group!r}")
    if group is not None and not 1 <= group <= 8:
        raise ValueError(f"group / PAR_GROUP must be an integer 1..8, got {group!r}")
    G_max = int(group) if group is not None else 8
    def lookahead(g):
        # a group's K_sinc is queued only once the NEXT group's items have been taken as well: whatever their producer queues on
        # the main stream then precedes the long launch and their plans run under it (with g - 1 extra items only, half of the
        # next group's plans started after the launch they should have hidden under: archive 165 -> 158 G, r06)
        return P if g <= 1 else max(P + g - 1, 2 * g)
    n_slots = 2 * lookahead(G_max)
    main = torch.cuda.current_stream(dev)
    ring = _borrow_plan_ring(dev, n_slots, P)
    sides, work, aux, free = ring["sides"], ring["work"], ring["aux"], ring["free"]

    def plan_item(item, j, ready):
        slot, stream = j % n_slots, sides[j % P]
        st_t, sp_t, sig_t = item[0], item[1], item[2]
        if sig_t.ndim == 2:
            len_in = sig_t.shape[0]
        else:
            len_in = item[4] if len(item) > 4 else sig_t.numel() // (item[3] if len(item) > 3 else 1)
        stream.wait_event(ready)                       # the item's tensors were produced on the main stream
        if free[slot] is not None:
            stream.wait_event(free[slot])              # the K_sinc that last read this slot is done
        # sizing reductions, (re)allocation and the plan itself all live on the side stream: a .item() there
        # does not wait for the K_sinc running on the main stream
        with torch.cuda.stream(stream):
            plan = speed_plan_dev(st_t, sp_t, len_in, dev, fused=True, work=work[slot], aux=aux[slot], stream=stream)
        # allocated (or regrown) under the side stream, read by the K_sinc on the main stream: without this a
        # block freed at generator teardown returns to the side stream's pool while that kernel still runs
        plan.work.record_stream(main)
        plan.aux.record_stream(main)
        work[slot], aux[slot] = plan.work, plan.aux        # keep (possibly grown) buffers for reuse
        return plan

    it = iter(items)
    ahead = collections.deque()                          # (index, item, future of its plan), oldest first
    pool = ThreadPoolExecutor(max_workers=P)
    try:
        j, k, done, g_now = 0, 0, False, 1
        while True:
            # take items BEFORE launching the next K_sinc: whatever their producer enqueues on the main stream (uploads,
            # generators) then precedes the long kernel instead of queueing behind it
            while not done and len(ahead) < lookahead(g_now):
                try:
                    item = next(it)
                except StopIteration:
                    done = True
                    break
                ready = torch.cuda.Event()
                ready.record(main)
                ahead.append((j, item, pool.submit(plan_item, item, j, ready)))
                j += 1
            if not ahead:
                return
            head = ahead[0][1]
            cls = _group_class(head)
            g_now = 1 if cls is None else (int(group) if group is not None else _group_size(head))
            if g_now > 1 and not done and len(ahead) < g_now:
                continue                                   # (take the rest of the group first)
            grp = [ahead.popleft()]
            while len(grp) < g_now and ahead and _group_class(ahead[0][1]) == cls:
                grp.append(ahead.popleft())
            plans = [fut.result() for _, _, fut in grp]
            if len(grp) > 1 and all(p.fused_ok for p in plans) and int(NT) == 32:
                outs = _resample_group(plans, [g_item for _, g_item, _ in grp], NT, dev)
            else:
                outs = [_resample_item(p, g_item, NT, dev) for p, (_, g_item, _) in zip(plans, grp)]
            ev = torch.cuda.Event()
            ev.record(main)
            for (k, _, _), out_t, plan in zip(grp, outs, plans):
                free[k % n_slots] = ev
                yield k, out_t, plan
    finally:
        for _, _, fut in ahead:
            fut.cancel()
        pool.shutdown(wait=True)
        ring["busy"] = False


# pinned staging / output slots of varispeed_batch_host, kept between calls (page-locking 0.5 GB costs ~0.1 s)
_pinned_ring = {}


def varispeed_batch_host(items, NT, dev=None):
    """Fused resampling of a batch of HOST-resident files on one GPU with the bus kept busy in both directions: the
    upload of file k+1 (its own stream) runs while file k's output is still going back (a third stream); plan and
    K_sinc (2-7 ms per 10-min file) sit between them on the current stream.  PCIe, not the kernels, bounds this form
    (DESIGN 5): it moves ~2x the samples per second of the upload -> compute -> download sequence.

    items: iterable of (sampletimes, speeds, signal): float64 numpy curves in samples / speed factors, signal a
    float32 numpy array or CPU tensor of shape (n,) or (n, ch) (interleaved, like soundfile returns it).  Pinned
    tensors are uploaded in place; anything else is staged through a pinned ring slot first (a host memcpy).
    Yields (index, out) in order, `out` a PINNED float32 CPU tensor (len_out,) or (len_out, ch) that stays valid until
    the generator is advanced again (two output slots alternate; the slots are kept for the next call, so one batch
    at a time per device)."""
    dev = _dev.device_index(dev)
    device = torch.device("cuda", dev)
    main = torch.cuda.current_stream(dev)
    up, down = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    pin_in, pin_out = _pinned_ring.setdefault(("in", dev), [None, None]), _pinned_ring.setdefault(("out", dev), [None, None])
    in_done, out_done = [None, None], [None, None]      # events: upload from / download into the slot finished
    work = aux = None
    pending = None                                       # (index, view of pin_out, event) of the previous file

    def pinned(buf, numel):
        if buf is None or buf.numel() < numel:
            buf = torch.empty(int(numel * 1.05) + 4096, dtype=torch.float32).pin_memory()
        return buf

    for k, (st, sp, sig) in enumerate(items):
        slot = k % 2
        st = np.ascontiguousarray(st, dtype=np.float64)
        sp = np.ascontiguousarray(sp, dtype=np.float64)
        if len(st) != len(sp) or len(st) < 2:
            raise ValueError("sampletimes and speeds must have the same length >= 2")
        src = sig if isinstance(sig, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(sig, dtype=np.float32))
        if src.dtype != torch.float32 or src.ndim not in (1, 2):
            raise ValueError("signal must be float32 of shape (n,) or (n, channels)")
        src = src.contiguous()
        if not src.is_pinned():
            if in_done[slot] is not None:
                in_done[slot].synchronize()              # the upload that last read this staging slot is through
            pin_in[slot] = pinned(pin_in[slot], src.numel())
            stage = pin_in[slot][:src.numel()].view(src.shape)
            _dev.host_copy(stage, src)                   # (on the staging threads: one core's memcpy was this loop's pace, r06)
            src = stage
        with torch.cuda.stream(up):
            st_t = torch.from_numpy(st).to(device, non_blocking=False)
            sp_t = torch.from_numpy(sp).to(device, non_blocking=False)
            sig_t = src.to(device, non_blocking=True)
            in_done[slot] = torch.cuda.Event()
            in_done[slot].record(up)
        main.wait_event(in_done[slot])
        # buffer bound from the host copy of the curve (fused_max_out's formula without a device read-back)
        max_out = int(float(sp.mean()) * float(st[-1] - st[0]) * 1.01) + int(float(np.diff(st).max()) * float(sp.max())) + 1024
        plan = speed_plan_dev(st_t, sp_t, src.shape[0], dev, fused=True, max_out=max_out, work=work, aux=aux)
        work, aux = plan.work, plan.aux
        out_t = _resample_item(plan, (st_t, sp_t, sig_t), NT, dev)
        done = torch.cuda.Event()
        done.record(main)
        for t in (st_t, sp_t, sig_t):                    # allocated on `up`, read on `main`: the caching allocator must
            t.record_stream(main)                        # not recycle them under the kernels
        pin_out[slot] = pinned(pin_out[slot], out_t.numel())     # slot of file k-2: the caller let go of it by advancing
        view = pin_out[slot][:out_t.numel()].view(out_t.shape)
        down.wait_event(done)
        with torch.cuda.stream(down):
            view.copy_(out_t, non_blocking=True)
            out_done[slot] = torch.cuda.Event()
            out_done[slot].record(down)
        out_t.record_stream(down)
        if pending is not None:                          # hand out file k-1 while file k is on the device / the bus
            pending[2].synchronize()
            yield pending[0], pending[1]
        pending = (k, view, out_done[slot])
    if pending is not None:
        pending[2].synchronize()
        yield pending[0], pending[1]


def varispeed_batch_gather(items, NT, dev=None, slots=3):
    """varispeed_batch_dev with the results GATHERED ON THE HOST (north_star; SURVEY 8e "results D2H into pinned host
    buffers"): inputs are device-resident work items as in varispeed_batch_dev, every output goes back over the bus into
    a ring of `slots` pinned buffers on its own stream while the next files' plans and K_sinc launches run.  This is the
    per-GPU loop of the end-to-end leg of the multi-GPU benchmark: the bus (one 0.92 GB D2H per 10-min stereo file), not
    the kernels, sets its pace.

    Yields (index, host_view, plan) in order; host_view is a PINNED float32 CPU tensor shaped like the device output,
    valid until `slots - 1` more items have been taken (the ring wraps)."""
    dev = _dev.device_index(dev)
    main = torch.cuda.current_stream(dev)
    down = torch.cuda.Stream(device=dev)
    ring = _pinned_ring.setdefault(("gather", dev), [None] * slots)
    while len(ring) < slots:
        ring.append(None)
    landed = [None] * slots                              # event: the D2H into slot s is complete
    pending = []                                         # (index, view, plan, event), oldest first
    for k, out_t, plan in varispeed_batch_dev(items, NT, dev):
        slot = k % slots
        if ring[slot] is None or ring[slot].numel() < out_t.numel():
            ring[slot] = torch.empty(int(out_t.numel() * 1.03) + 4096, dtype=torch.float32).pin_memory()
        view = ring[slot][:out_t.numel()].view(out_t.shape)
        done = torch.cuda.Event()
        done.record(main)
        down.wait_event(done)
        with torch.cuda.stream(down):
            view.copy_(out_t, non_blocking=True)
            landed[slot] = torch.cuda.Event()
            landed[slot].record(down)
        out_t.record_stream(down)
        pending.append((k, view, plan, landed[slot]))
        if len(pending) >= slots - 1:                    # keep slots - 1 copies in flight behind the compute
            i, v, p, ev = pending.pop(0)
            ev.synchronize()
            yield i, v, p
    for i, v, p, ev in pending:
        ev.synchronize()
        yield i, v, p


def varispeed_resample_dev(plan, sig_t, NT, out_t=None, pos_t=None, sig_stride=1, len_in=None, out_stride=1, n_chunks=0):
    """Positions + sinc interpolation of one channel from a SpeedPlan (position array materialised in
    `pos_t`, optionally pipelined in chunks).  Returns (out, pos)."""
    dev = plan.dev
    L = _lib.lib()
    if len_in is None:
        len_in = sig_t.numel() // sig_stride
    if pos_t is None:
        pos_t = _dev.empty(plan.len_out, torch.float64, dev)
    if out_t is None:
        out_t = _dev.empty(plan.len_out * out_stride, torch.float32, dev)
    _lib.check(L.par_varispeed_resample_f32(dev, _dev.ptr(plan.speeds_t), plan.m, _dev.ptr(plan.work), plan.len_out,
                                            _dev.ptr(pos_t), _dev.ptr(sig_t), sig_stride, len_in, int(NT), _dev.ptr(out_t),
                                            out_stride, n_chunks, _dev.stream_ptr(dev)))
    return out_t, pos_t


def speed_to_pos(sampletimes, speeds, num_imput_samples):
    """
    sampletimes: 1D array of sample numbers at which speeds is sampled; must have even spacing
    speeds: 1D array of speed samples
    num_imput_samples: int
    Returns the float64 read positions (bit-identical to the reference's array; when the reference's
    end trim does not fire its buffer tail is uninitialised -- only the written prefix is returned).
    """
    dev = _dev.device_index(None)
    st = _dev.to_dev(np.asarray(sampletimes, dtype=np.float64), torch.float64, dev)
    sp = _dev.to_dev(np.asarray(speeds, dtype=np.float64), torch.float64, dev)
    return _dev.to_host(speed_to_pos_dev(st, sp, num_imput_samples, dev))


# ----------------------------------------------------------------------------- interpolation

def lag_to_pos_dev(lag_curve, sr, num_input_samples, dev=None):
    """Lag-curve branch of resampling.run (util/resampling.py:189-206) on the device: np.interp over
    arange(num_output_samples), find_cutoff trim and clip(0) in one kernel; the float64 position array is
    born in HBM.  lag_curve: (m, 2) array of (time s, lag s).  -> device f64 positions (trimmed view)."""
    dev = _dev.device_index(dev)
    lag_curve = np.asarray(lag_curve, dtype=np.float64)
    sampletimes = lag_curve[:, 0] * sr
    lags = lag_curve[:, 1] * sr
    num_output_samples = num_input_samples + abs(lags[-1])
    num_out = int(np.ceil(num_output_samples))                        # len(np.arange(float stop))
    xp_t = _dev.to_dev(sampletimes, torch.float64, dev)
    fp_t = _dev.to_dev(sampletimes - lags, torch.float64, dev)
    pos_t = _dev.empty(num_out, torch.float64, dev)
    work = _dev.empty(1, torch.int64, dev)
    len_out, trimmed = ctypes.c_int64(0), ctypes.c_int(0)
    _lib.check(_lib.lib().par_lag_to_pos_f64(dev, _dev.ptr(xp_t), _dev.ptr(fp_t), len(sampletimes), num_out,
                                             int(num_input_samples), _dev.ptr(pos_t), _dev.ptr(work),
                                             ctypes.byref(len_out), ctypes.byref(trimmed), _dev.stream_ptr(dev)))
    if trimmed.value:
        logging.debug(f"Trimmed to sample {len_out.value}")
    return pos_t[:len_out.value]


def sinc_resample_dev(pos_t, sig_t, NT, out_t=None, sig_stride=1, len_in=None, out_stride=1, dev=None):
    """pos_t float64[len_out], sig_t float32 (stride sig_stride) -> float32 out (stride out_stride)."""
    dev = _dev.device_index(dev if dev is not None else pos_t.device)
    L = _lib.lib()
    len_out = pos_t.numel()
    if len_in is None:
        len_in = sig_t.numel() // sig_stride
    if out_t is None:
        out_t = _dev.empty(len_out * out_stride, torch.float32, dev)
    if len_out == 0:
        return out_t
    _lib.check(L.par_sinc_resample_f32(dev, _dev.ptr(pos_t), len_out, _dev.ptr(sig_t), sig_stride, len_in, int(NT),
                                       _dev.ptr(out_t), out_stride, _dev.stream_ptr(dev)))
    return out_t


def linear_resample_dev(pos_t, sig_t, out_t=None, sig_stride=1, len_in=None, out_stride=1, dev=None):
    dev = _dev.device_index(dev if dev is not None else pos_t.device)
    L = _lib.lib()
    len_out = pos_t.numel()
    if len_in is None:
        len_in = sig_t.numel() // sig_stride
    if out_t is None:
        out_t = _dev.empty(len_out * out_stride, torch.float32, dev)
    _lib.check(L.par_linear_resample_f32(dev, _dev.ptr(pos_t), len_out, _dev.ptr(sig_t), sig_stride, len_in,
                                         _dev.ptr(out_t), out_stride, _dev.stream_ptr(dev)))
    return out_t


def sinc_wrapper(sample_at, signal, lowpass, NT):
    """Returns float32[len(sample_at)].  `lowpass` is ignored, as in the reference (dead argument)."""
    dev = _dev.device_index(None)
    if len(sample_at) == 0:                         # the reference's loop never runs: an empty float32 array
        return np.empty(0, np.float32)
    if len(sample_at) == 1:
        raise UnboundLocalError("local variable 'period_to' referenced before assignment")   # reference behaviour
    pos_t = _dev.to_dev(np.asarray(sample_at, dtype=np.float64), torch.float64, dev)
    # the reference's int(round(p)) raises on non-finite positions -- looked for in HBM (a host pass over 115 M positions was
    # 40 % of this call, r06)
    if not bool(torch.isfinite(pos_t).all()):
        if bool(torch.isnan(pos_t).any()):
            raise ValueError("cannot convert float NaN to integer")
        raise OverflowError("cannot convert float infinity to integer")
    sig_t = _dev.to_dev(signal, torch.float32, dev)
    return _dev.to_host(sinc_resample_dev(pos_t, sig_t, NT, dev=dev))


def sinc_wrapper_mt(output, sample_at, signal, lowpass, NT):
    """In-place variant: fills caller-owned `output` (may be a strided column view), returns None.
    Uses the canonical period definition (every chunk boundary reads the true next position), so the
    result does not depend on os.cpu_count() the way the reference's chunked threads do."""
    _dev.host_assign(output, sinc_wrapper(sample_at, signal, lowpass, NT))      # (a column view: scattered on the staging threads)


def sinc_core(sample_at, signal, lowpass, output, win_func, N):
    """Reference-signature entry (util/resampling.py:52): NT from len(N); window/N are recomputed on
    the device side from NT exactly as sinc_wrapper builds them."""
    NT = (len(N) - 1) // 2
    _dev.host_assign(output, sinc_wrapper(sample_at, signal, lowpass, NT))


# ----------------------------------------------------------------------------- run

class _Progress:
    """Progress reporting with the reference's signal protocol: `prog_sig.notifyProgress.emit(value)`; silent
    when no signal object was given."""

    def __init__(self, prog_sig):
        self._emit = prog_sig.notifyProgress.emit if prog_sig else (lambda value: None)

    def __call__(self, value):
        self._emit(value)


def _plan_positions(sig_shape, sr, speed_curve, lag_curve, want_fused, dev):
    """(SpeedPlan or None, float64 position tensor or None) for one file.  Speed curve: the plan; its position
    array is only materialised when the fused K_sinc cannot be used.  Lag curve: K_lag positions."""
    n_in = sig_shape[0]
    if speed_curve is not None:
        curve = np.asarray(speed_curve, dtype=np.float64)
        st_t = _dev.to_dev(curve[:, 0] * sr, torch.float64, dev)
        sp_t = _dev.to_dev(np.ascontiguousarray(curve[:, 1]), torch.float64, dev)
        if want_fused:
            plan = speed_plan_dev(st_t, sp_t, n_in, dev, fused=True)
            if plan.fused_ok:
                return plan, None
        return None, speed_to_pos_dev(st_t, sp_t, n_in, dev)
    if lag_curve is not None:
        return None, lag_to_pos_dev(lag_curve, sr, n_in, dev)
    # the reference reaches its channel loop with `sample_at` never assigned
    raise UnboundLocalError("local variable 'sample_at' referenced before assignment")


def run(filenames, signal_data=None, speed_curve=None, resampling_mode="Linear", sinc_quality=50, use_channels=(),
        prog_sig=None, lag_curve=None, suffix=""):
    """Batch resampler with the contract of the reference's resampling.run (util/resampling.py:162-240): per file,
    positions from the speed curve (or lag curve), every selected channel resampled ("Sinc" or "Linear"), result
    written as `<stem>_res<suffix>.wav` (32-bit float), progress emitted as 0, then (k+1)/channels*100 per
    channel, then 100 per file.  `signal_data` optionally supplies decoded `(signal, sr)` pairs.  The file's
    channels, its positions (or the fused plan) and the interleaved output stay in HBM; one D2H per file."""
    from . import io_ops
    progress = _Progress(prog_sig)
    progress(0)
    dev = _dev.device_index(None)
    decoded = signal_data if signal_data is not None else [None] * len(filenames)
    for filename, given in zip(filenames, decoded):
        with log_duration("Preparing"):
            logging.info(f"Resampling '{os.path.basename(filename)}'... {resampling_mode}, {sinc_quality}, {use_channels}")
            signal, sr = given if given else io_ops.read_file(filename)[:2]
            n_in, n_ch_in = signal.shape
            sig_t = _dev.to_dev(signal, torch.float32, dev)                  # (n, ch) interleaved, like the file
            plan, pos_t = _plan_positions(signal.shape, sr, speed_curve, lag_curve, resampling_mode == "Sinc", dev)
        # channel selection persists across files, as the reference rebinds its argument
        use_channels = [c for c in use_channels if c < n_ch_in] if use_channels else tuple(range(n_ch_in))
        with log_duration("Resampling"):
            n_out_ch = len(use_channels)
            length = plan.len_out if pos_t is None else pos_t.numel()
            out_t = _dev.empty((length, n_out_ch), torch.float32, dev)
            layout = dict(sig_stride=n_ch_in, len_in=n_in, out_stride=n_out_ch)
            flat_in, flat_out = sig_t.reshape(-1), out_t.reshape(-1)                # strided channel views, no copies
            k = 0
            while k < n_out_ch:
                ch = use_channels[k]
                if resampling_mode == "Sinc" and pos_t is None and k + 1 < n_out_ch:
                    # two channels per launch: they share the positions, hence the whole weight computation
                    varispeed_fused_stereo_dev(plan, flat_in[ch:], flat_in[use_channels[k + 1]:], sinc_quality,
                                               flat_out[k:], flat_out[k + 1:], **layout)
                    progress((k + 1) / n_out_ch * 100)
                    k += 1
                elif resampling_mode == "Sinc":
                    if pos_t is None:
                        varispeed_fused_dev(plan, flat_in[ch:], sinc_quality, flat_out[k:], **layout)
                    else:
                        sinc_resample_dev(pos_t, flat_in[ch:], sinc_quality, flat_out[k:], dev=dev, **layout)
                elif resampling_mode == "Linear":
                    linear_resample_dev(pos_t, flat_in[ch:], flat_out[k:], dev=dev, **layout)
                progress((k + 1) / n_out_ch * 100)
                k += 1
            # pinned staging for the one D2H of the file (4x the pageable rate, tools/bench_e2e.py)
            host = torch.empty(out_t.shape, dtype=torch.float32, pin_memory=True)
            host.copy_(out_t)
            result = host.numpy()
        with log_duration("Writing"):
            io_ops.write_wav_float(f"{os.path.splitext(filename)[0]}_res{suffix}.wav", result, sr)
            progress(100)
    logging.info("Done!")
