#!/usr/bin/env python3
"""bench.py -- north-star benchmark: Msamples/s resampled, 192 kHz varispeed, 64-tap sinc.

One "step" = one pass of the varispeed hot path over one file that is already resident in HBM:
speed curve -> segment plan (lazy: closed-form segment sums, block records; csrc/pos_plan.h) -> Hann-windowed sinc
interpolation with the outputs placed from those records inside K_sinc (no position array; --no-fused materialises the
reference's float64 sample_at) -> float32 output in HBM.  Workload (BASELINE.json configs[1] at the metric's 192 kHz):
60-min mono float32, speed 1 + 0.01 sin(2 pi 0.55 t + 0.7) sampled every 256 samples, NT = 32
(SURVEY 8d).  Inputs are synthesised on the device (closed form + stateless hash noise).

N > 1 (or --config5): BASELINE config 5 -- the 512-file archive (192 kHz stereo, 10 min each) as ONE step,
shared out over the ranks through a host-side work queue (files/channels are independent in the reference,
util/resampling.py:168,225): one process per GPU, no data-path collective, no RCCL (gloo barrier + MAX/SUM of
time and sample counts).  Strong scaling: the batch is fixed, `value` = aggregate channel-samples/s; the line also
carries the SAME workload on one GPU measured in the same run (n1_same_workload_value, speedup_vs_n1, efficiency)
and the host-gather leg (value_e2e: every output copied to pinned host memory).
`python bench.py --gpus N` without a launcher starts its own N ranks (torch.distributed.run, gloo, 127.0.0.1);
under the driver's `python -m torch.distributed.run ... bench.py --gpus N` it joins the ranks it was given.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HIP-event-timed
K_sinc launches vs the 8 TB/s HBM peak, 8 algorithmic bytes per output sample) and `cpu_baseline`
(the plain-C oracle port timed on this box's host cores on a bounded sample), and `parity_checked`: windows of the
TIMED output buffer compared with the oracle outside the timed region.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALGO_BYTES_PER_SAMPLE = 8.0    # 4 B signal read + 4 B output write


def usable_cores():
    """(cores this process may run on, cores the box reports): the scheduler affinity mask, further limited by the cgroup's CPU
    quota when there is one (cpu.max / cfs_quota_us); os.cpu_count() alone reports the HOST's cores inside a limited container
    ("cores: 256" for a rate worth ~17 of them)."""
    reported = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = reported
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, reported


def cpu_baseline(sr, nt, budget_s=15.0):
    """Oracle C port (oracle/par_oracle.c) on the host cores: speed_to_pos (1 thread, serial like the
    reference) + sinc_mt (one contiguous chunk per usable core, like sinc_wrapper_mt's one per os.cpu_count(),
    util/resampling.py:33-46)."""
    import numpy as np
    from oracle import oracle_c as C
    cores, reported = usable_cores()

    def run(seconds, threads):
        n = int(sr * seconds)
        sig = C.synth_signal(0, n, float(sr))
        m = int(seconds * sr / 256)
        st, sp = C.synth_curve(m, seconds, float(sr))
        t0 = time.perf_counter()
        pos, _ = C.speed_to_pos(st, sp, n)
        tp = time.perf_counter() - t0
        out = C.sinc(pos, sig, nt, threads=threads)
        dt = time.perf_counter() - t0
        return len(out), dt, tp

    n1, t1, tp1 = run(0.5, 1)                        # probe = the per-core rate of the interpolator (one thread)
    per_core = n1 / max(t1 - tp1, 1e-9)
    use = cores
    rate = per_core * min(cores, 8)                  # sizes the sample only (a conservative guess of the N-thread rate)
    seconds = max(1.0, min(600.0, budget_s * rate / sr))
    n2, t2, tp2 = run(seconds, use)
    sinc_rate = n2 / max(t2 - tp2, 1e-9)
    return {"value": round(n2 / t2 / 1e6, 3), "unit": "Msamples/s", "cores": use, "cores_usable": cores, "cores_reported": reported,
            "kind": "port", "per_core_Msamples/s": round(per_core / 1e6, 3),
            "sinc_only_value": round(sinc_rate / 1e6, 3), "sinc_only_parallel_efficiency": round(sinc_rate / (per_core * use), 3),
            "speed_to_pos_share": round(tp2 / t2, 3),
            "sample": f"{seconds:.1f} s of the same 192 kHz workload ({n2} output samples, {t2:.1f} s wall): "
                      f"C speed_to_pos on 1 thread ({100 * tp2 / t2:.0f} % of the wall time: the reference's own serial loop, "
                      f"which caps this line) + C sinc on {use} threads = the cores this process may use (affinity mask and cgroup "
                      f"quota; the box reports {reported}), contiguous chunks like sinc_wrapper_mt; sinc_only_value = the interpolator "
                      f"alone, per_core_Msamples/s = the same on ONE thread (0.5-s probe)"}


def parity_windows(out, sig, st, spd, n_in, n_out, nt, n_windows=16, width=2000):
    """The checker's leg, OUTSIDE the timed region: windows of the TIMED output buffer against the oracle -- positions by the
    oracle's own speed_to_pos arithmetic (oracle_speed_to_pos_windows: the reference's chain, segment sums on the usable cores),
    samples by oracle_c.sinc on the signal's stretch under each window.  Windows: spread over the file, its first and last
    tiles, the launch's short tail streams.  Returns the `parity_checked` object of the bench line."""
    import numpy as np
    from oracle import oracle_c as C
    cores, _ = usable_cores()
    t0 = time.perf_counter()
    last = n_out - width - 1
    starts = sorted({int(v) for v in np.linspace(0, last, n_windows - 4)} | {0, last, max(0, last - 1024), max(0, last - 2048 * 24 * 1024 + 4096)})
    st_h, sp_h = st.cpu().numpy(), spd.cpu().numpy()
    pos, len_ref, _ = C.speed_to_pos_windows(st_h, sp_h, n_in, starts, width + 1, threads=cores)
    assert len_ref == n_out, f"len_out {n_out} != the oracle's {len_ref}"
    worst, worst_at = 0.0, -1
    for w, i in enumerate(starts):
        p = pos[w]
        assert not np.isnan(p).any()
        lo = max(0, int(p[0]) - 200)
        hi = min(n_in, int(p[-1]) + 200)
        ref = C.sinc(p - lo, sig[lo:hi].cpu().numpy(), nt)[:width]
        got = out[i:i + width].cpu().numpy()
        e = float(np.max(np.abs(got - ref)) / max(float(np.max(np.abs(ref))), 1e-30))
        if e > worst:
            worst, worst_at = e, i
    assert worst < 1e-5, f"timed output differs from the oracle: {worst:.3e} in the window at output {worst_at}"
    return {"windows": len(starts), "outputs_per_window": width, "max_rel": float(f"{worst:.3e}"), "tolerance": 1e-5,
            "len_out_equals_oracle": True, "seconds": round(time.perf_counter() - t0, 2),
            "what": "windows of the timed output buffer (last timed step) against oracle_c.sinc at the oracle's own positions "
                    "(oracle_speed_to_pos_windows: same arithmetic as the reference's speed_to_pos, pinned bit for bit in "
                    "tests/test_oracle_golden.py); spread over the file + first / last tiles + the launch's short tail streams; "
                    "the whole file's positions and 36+ windows: tests/test_hip_parity.py::test_full_size_benchmark_workload",
            "p0_note": "the config-3 end-to-end chain (pyrespeeder on flutter_192.flac) is held to a COUNT bound, not to 1e-5 on every "
                       "sample: 0 of 101 435 dense samples beyond 1e-5, 6 of 9 000 samples of the end window up to 1.42e-5 (the fixture's "
                       "own float32 FFT noise summed by the position cumsum; DESIGN section 2) -- stage-wise 1e-5 holds"}


def stft_secondary(sig, dev, n_fft=1024, hop=256, cpu=True):
    """Secondary line (not the metric): K_stft magnitude throughput on the same resident signal, against
    its 12.02 B/sample HBM roofline and the reference's own GPU route torch.stft + abs
    (util/fourier.py:101-107, rocFFT) -- a reported baseline, not the optimisation target."""
    import math
    import numpy as np
    import scipy.signal
    import torch
    from pyaudiorestoration_amd import fourier
    x = sig[:min(sig.numel(), 57_600_000)]
    n = x.numel()
    win = torch.from_numpy(scipy.signal.get_window("blackmanharris", n_fft).astype(np.float32)).to(x.device)

    def t_of(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    t_mag = t_of(lambda: fourier.stft_dev(x, n_fft, hop, win, 1, 1, dev=dev))

    def ref():
        r = torch.stft(x, n_fft, hop_length=hop, window=win, win_length=n_fft, center=True, pad_mode="reflect",
                       normalized=False, onesided=True, return_complex=True)
        r /= math.sqrt(n_fft)
        return r.abs() + 1e-7
    t_ref = t_of(ref, 3)
    bps = 4 + 4 * (n_fft / 2 + 1) / hop
    res = {"kernel": "k_stft get_mag 1024/256", "samples": n, "Msamples/s": round(n / t_mag / 1e6, 1),
           "algorithmic_GB/s": round(n * bps / t_mag / 1e9, 1), "frac_of_hbm_peak": round(n * bps / t_mag / 8e12, 4),
           "torch_stft_abs_Msamples/s": round(n / t_ref / 1e6, 1)}
    if cpu:
        # CPU lines of BASELINE.md section 4 (reported baselines, bounded samples): (a) the reference's numpy route --
        # np.fft.rfft over reflect-padded windowed frames, 1 thread (oracle_np restates util/fourier.py:136-157);
        # (b) the C port, frames split over the host cores
        from oracle import oracle_c as C
        from oracle import oracle_np as O
        xs = x[:min(n, 20 * 192000)].cpu().numpy()
        wn = win.cpu().numpy()
        t0 = time.perf_counter()
        O.get_mag(xs[:4 * 192000], n_fft, hop, "blackmanharris")
        t_np = time.perf_counter() - t0
        res["numpy_rfft_1thread_Msamples/s"] = round(4 * 192000 / t_np / 1e6, 2)
        cores, _ = usable_cores()                   # (the cores this process may use, like cpu_baseline: not the host's 256)
        t0 = time.perf_counter()
        C.stft(xs, n_fft, hop, wn, 1, mode=1, threads=cores)
        t_c = time.perf_counter() - t0
        res["c_port_Msamples/s"] = round(len(xs) / t_c / 1e6, 2)
        res["c_port_threads"] = cores
        res["cpu_sample"] = f"numpy: 4 s of audio, 1 thread; C port: {len(xs) / 192000:g} s of audio, {cores} threads (frames split)"
    return res


def heal_secondary(dev, tiles=256):
    """Secondary line: BASELINE config 4 shape -- 322 531-sample signal tiled x256, STFT 512/32 -> 32 inpaint boxes
    per tile (one K_heal launch) -> apply-and-clear -> fused ISTFT, resident in HBM; 136.5 algorithmic B/sample
    (SURVEY 8d).  Synthetic stand-in with dropouts_sample.flac's length and rate."""
    import numpy as np
    import scipy.signal
    import torch
    from pyaudiorestoration_amd import _dev, _lib, pipeline
    L = _lib.lib()
    n1, sr, n_fft, hop = 322531, 44100, 512, 32
    n = n1 * tiles
    s = _dev.stream_ptr(dev)
    x = torch.zeros(n + n_fft // 2, dtype=torch.float32, device=f"cuda:{dev}")
    _lib.check(L.par_synth_signal_f32(dev, _dev.ptr(x), 0, n, float(sr), 0x5EED, s))
    rng = np.random.default_rng(4)
    marks = []
    for k in range(tiles):
        for t in np.sort(rng.uniform(0.2, n1 / sr - 0.2, 32)):
            w = rng.uniform(0.004, 0.02)
            marks.append((k * n1 / sr + t - w / 2, 500.0, k * n1 / sr + t + w / 2, 9000.0, 0.5))
    geo = torch.from_numpy(np.array([pipeline.marker_geometry(m, sr, hop, n_fft) for m in marks], dtype=np.int32)).to(x.device)
    win = torch.from_numpy(scipy.signal.get_window("blackmanharris", n_fft).astype(np.float32)).to(x.device)
    frames, bins = int(L.par_stft_frames(n + n_fft // 2, n_fft, hop)), n_fft // 2 + 1
    spec = torch.empty((frames, bins), dtype=torch.complex64, device=x.device)
    gain = torch.zeros((frames, bins), dtype=torch.float32, device=x.device)
    y = torch.empty(n, dtype=torch.float32, device=x.device)

    def step():
        _lib.check(L.par_stft_f32(dev, _dev.ptr(x), n + n_fft // 2, 1, n_fft, hop, 1, _dev.ptr(win), _dev.ptr(spec), 0, 0, s))
        _lib.check(L.par_inpaint_gain_db_c64(dev, _dev.ptr(spec), frames, bins, _dev.ptr(geo), len(marks), _dev.ptr(gain), s))
        _lib.check(L.par_spec_apply_gain_boxes_c64(dev, _dev.ptr(spec), frames, bins, _dev.ptr(geo), len(marks), _dev.ptr(gain), s))
        _lib.check(L.par_istft_f32(dev, _dev.ptr(spec), frames, n_fft, hop, _dev.ptr(win), None, _dev.ptr(y), n, n_fft // 2, s))
    # (this leg follows the CPU legs of the STFT line, seconds of host-only work: 100 ms of untimed steps bring the clocks back)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        step()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    dense = {"ms": round(dt * 1e3, 3), "Msamples/s": round(n / dt / 1e6, 1), "algorithmic_GB/s": round(n * 136.5 / dt / 1e9, 1),
             "frac_of_hbm_peak": round(n * 136.5 / dt / 8e12, 4),
             "what": "the reference-shaped DENSE chain (transform, heal and invert the whole signal: 136.5 B/sample, two passes over "
                     "a 5.3 GB c64 spectrogram) -- kept as the cross-check of the sparse path (equal to 2e-6), RETIRED as a "
                     "roofline target in r06 (DESIGN 0, item 7): the product path below moves 27 % of those bytes"}
    res = {"workload": f"config 4: dropout inpaint, {n1} samples x{tiles} tiles, stft 512/32, {len(marks)} boxes", "samples": n,
           "ms": dense["ms"], "Msamples/s": dense["Msamples/s"], "path": "dense", "dense_reference_shape": dense}
    # r03 sparse path (pipeline.heal_dropouts' default): only the frames a box can reach are transformed, the rest of the
    # signal is copied -- host plan (numpy merge of the 8192 boxes' frame ranges), gather, STFT, inpaint, ISTFT, scatter
    del spec, gain
    sig2 = x[:n].reshape(n, 1)
    out2 = torch.empty((n, 1), dtype=torch.float32, device=x.device)
    geometry = geo.cpu().numpy().astype(np.int64)
    t0 = time.perf_counter()
    plan = pipeline.heal_segments(geometry, (n + n_fft // 2) // hop + 1, n + n_fft // 2, n, n_fft, hop)
    t_plan = time.perf_counter() - t0
    if plan is not None:
        g2 = None
        g2, _ = pipeline.heal_dropouts_dev(sig2, n, 1, 0, geometry, n_fft, hop, out2, dev, True, g2, plan)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g2, _ = pipeline.heal_dropouts_dev(sig2, n, 1, 0, geometry, n_fft, hop, out2, dev, True, g2, plan)
        torch.cuda.synchronize()
        dts = (time.perf_counter() - t0) / 5
        # the line's own ms / Msamples/s are the PRODUCT path's (pipeline.heal_dropouts' default: sparse)
        res["ms"], res["Msamples/s"], res["path"] = round(dts * 1e3, 3), round(n / dts / 1e6, 1), "sparse (pipeline.heal_dropouts' default)"
        res["sparse"] = {"ms": round(dts * 1e3, 3), "Msamples/s": round(n / dts / 1e6, 1), "segments": plan["segments"],
                         "samples_transformed": plan["total"], "fraction_of_signal": round(plan["total"] / n, 4),
                         "host_plan_ms": round(t_plan * 1e3, 3),
                         "what": "wall time of pipeline.heal_dropouts_dev per call (gather, STFT, inpaint, ISTFT, copy + scatter) "
                                 "with the segment plan given -- the plan is a function of the markers, made once per file on the "
                                 "host (host_plan_ms); equals the dense result to 2e-6 "
                                 "(tests/test_hip_parity.py::test_config4_x256_tiles_one_launch)"}
    return res


def nt50_secondary(dev, sig, st, spd, n_in, work, aux, cap, n_out):
    """Secondary line: the timed file at the reference's DEFAULT quality (sinc_quality = 50, util/resampling.py:162): 100 taps.  The
    block kernel: fc = 1 taps as a Farrow bank on the matrix cores (four K slices, r06), fc < 1 taps on the vector units (no
    moment form for this tap count).  K_sinc alone, 6 launches back to back behind 2 warm ones, from the timed file's plan."""
    import torch
    from pyaudiorestoration_amd import _dev, _lib
    L = _lib.lib()
    s = _dev.stream_ptr(dev)
    m = st.numel()
    out = torch.empty(cap, dtype=torch.float32, device=f"cuda:{dev}")
    def launch():
        _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(spd), m, _dev.ptr(work), _dev.ptr(aux), cap, n_out, _dev.ptr(sig), 1, n_in, 50,
                                             _dev.ptr(out), 1, s))
    for _ in range(2):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream(dev))
    for _ in range(6):
        launch()
    e1.record(torch.cuda.current_stream(dev))
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 6
    return {"workload": "the timed file at NT = 50 (the reference's default sinc_quality): 100-tap Hann sinc", "ms": round(ms, 3),
            "Msamples/s": round(n_out / ms / 1e3, 1), "frac_of_hbm_peak": round(ALGO_BYTES_PER_SAMPLE * n_out / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "kernel": "k_sinc_fused<1, 50, 4> (block kernel): fc = 1 far taps as a Farrow bank on v_mfma_f32_16x16x32_f16, fc < 1 taps on the vector units"}


def stereo_secondary(dev, sr=192000, seconds=600.0, nt=32):
    """Secondary line: one work item of BASELINE config 5 -- a 10-min 192 kHz STEREO file, interleaved (n, 2) like the
    reference holds it: one plan per file and ONE stereo K_sinc launch (positions, prologue and tap weights shared by the
    two channels)."""
    import torch
    from pyaudiorestoration_amd import _dev, _lib
    L = _lib.lib()
    s = _dev.stream_ptr(dev)
    n = int(sr * seconds)
    m = int(seconds * sr / 256)
    sig = torch.empty((n, 2), dtype=torch.float32, device=f"cuda:{dev}")
    mono = torch.empty(n, dtype=torch.float32, device=f"cuda:{dev}")
    for c in range(2):
        _lib.check(L.par_synth_signal_f32(dev, _dev.ptr(mono), 0, n, float(sr), 0x5EED + c, s))
        sig[:, c] = mono
    del mono
    st = torch.empty(m, dtype=torch.float64, device=f"cuda:{dev}")
    sp = torch.empty(m, dtype=torch.float64, device=f"cuda:{dev}")
    _lib.check(L.par_synth_speed_curve_f64(dev, _dev.ptr(st), _dev.ptr(sp), m, seconds, float(sr), 0.01, 0.55, 0.7, s))
    cap = int(n * 1.02) + 1024
    nbytes, aux_bytes = int(L.par_speed_plan_bytes(m)), int(L.par_fused_aux_bytes(cap, m))
    work = torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{dev}")
    aux = torch.empty(aux_bytes, dtype=torch.uint8, device=f"cuda:{dev}")
    out = torch.empty((cap, 2), dtype=torch.float32, device=f"cuda:{dev}")
    len_out, trimmed, ok = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
    base_in, base_out = sig.data_ptr(), out.data_ptr()

    def step():
        _lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st), _dev.ptr(sp), m, n, _dev.ptr(work), nbytes, _dev.ptr(aux),
                                                 aux_bytes, cap, ctypes.byref(len_out), ctypes.byref(trimmed), 0, None,
                                                 ctypes.byref(ok), s))
        _lib.check(L.par_varispeed_fused_stereo_f32(dev, _dev.ptr(sp), m, _dev.ptr(work), _dev.ptr(aux), cap, len_out.value,
                                                    ctypes.c_void_p(base_in), ctypes.c_void_p(base_in + 4), 2, n, nt,
                                                    ctypes.c_void_p(base_out), ctypes.c_void_p(base_out + 4), 2, s))
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    # the same file as a batch of 48 through the library's batch driver (next file's plan on a side stream under K_sinc):
    # what one GPU does in the N > 1 / --config5 mode
    from pyaudiorestoration_amd import resampling
    del out, work, aux
    items = [(st, sp, sig)] * 48
    # warm-up: the whole batch once, untimed -- every plan slot's buffers exist afterwards, and the GPU is back at its working
    # clocks (this leg follows seconds of host-only work, the CPU legs of the other secondaries: timed cold, the 65 ms of
    # the batch read 5-8 % low -- tools/exp/archive_gap.py has the same call at 1.26-1.27 ms per file after 10 s of GPU work)
    for _ in resampling.varispeed_batch_dev(items, nt, dev=dev):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in resampling.varispeed_batch_dev(items, nt, dev=dev):
        pass
    torch.cuda.synchronize()
    dtb = (time.perf_counter() - t0) / len(items)
    return {"workload": f"config 5 work item: {seconds:g}-s {sr} Hz stereo file, interleaved; 1 plan + 1 stereo fused K_sinc launch",
            "channel_samples_out": 2 * len_out.value, "ms_per_file": round(dt * 1e3, 3),
            "Msamples/s": round(2 * len_out.value / dt / 1e6, 1),
            "batched_ms_per_file": round(dtb * 1e3, 3), "batched_Msamples/s": round(2 * len_out.value / dtb / 1e6, 1),
            "note": "batched_Msamples/s is the 1-GPU point of the --gpus N > 1 curve: those runs time the config-5 archive (this "
                    "work item x 512, files pulled from a shared queue), not the mono file of this line's `value`; the whole "
                    "archive on one GPU: python bench.py --config5 (profiles/r06_bench_config5_n1.json)"}


def _launch_ranks(a):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks ourselves (one
    process per GPU, gloo rendezvous on the loopback interface) by re-executing this file under torch.distributed.run,
    and pass rank 0's JSON line through.  Fewer than N visible devices is an error, not a fold onto what there is."""
    import socket
    import subprocess
    if not (a.dry_run or os.environ.get("PAR_OVERSUBSCRIBE") == "1"):
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < a.gpus:
            sys.exit(f"bench.py: --gpus {a.gpus} but {have} GPU(s) visible; refusing to fold ranks onto fewer devices "
                     "(PAR_OVERSUBSCRIBE=1 shares devices on purpose)")
    with socket.socket() as sk:                          # a free rendezvous port (the queue's store binds above it)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    if a.dry_run:
        env["PAR_OVERSUBSCRIBE"] = "1"                   # no GPU work in a dry run: the device count is irrelevant
    return subprocess.call(cmd, env=env)


def config5_dry_run(a, ctx):
    """--dry-run: the launch, rendezvous, work queue, same-workload base, reductions and the JSON line of the config-5
    mode with the GPU work replaced by a sleep per file -- what the CPU tests drive (no HIP call, `value` means nothing)."""
    from pyaudiorestoration_amd import multi_gpu
    per_file = 2 * 115_200_000
    done = {"files": 0}
    step_no = [0]

    def run_files(file_iter):
        k = 0
        for _ in file_iter:
            time.sleep(0.002)
            k += 1
        return k

    def step():
        q = multi_gpu.WorkQueue(ctx, range(a.files), f"d{step_no[0]}")
        step_no[0] += 1
        done["files"] = run_files(q)

    for _ in range(a.warmup):
        step()
    ctx.barrier()
    n1 = None
    if ctx.rank == 0:
        t0 = time.perf_counter()
        k = run_files(range(min(a.files, a.n1_files)))
        n1 = k * per_file / (time.perf_counter() - t0) / 1e6
    dt = ctx.timed(step, a.steps)
    total = ctx.reduce_sum(done["files"]) * per_file
    if ctx.rank == 0:
        value = total * a.steps / dt / 1e6
        print(json.dumps({"metric": "Msamples/sec resampled (192 kHz varispeed)", "value": round(value, 3), "unit": "Msamples/s",
                          "n_gpus": ctx.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dry_run": True,
                          "data": "none (dry run: a sleep per file)", "config": {"workload": "config 5 flow, dry run", "files": a.files,
                                                                                 "channel_samples_per_step": int(total)},
                          "n1_same_workload_value": round(n1, 3), "speedup_vs_n1": round(value / n1, 3),
                          "speedup_base": "n1_same_workload_value (the archive on one GPU, this run) -- the N = 1 line carries the "
                                          "same quantity as archive_value",
                          "efficiency": round(value / n1 / ctx.world, 4),
                          "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                                       "kernel_ms_per_file_alone_min_max": None},
                          "cpu_baseline": None,
                          "cpu_baseline_note": "timed on the N = 1 lines only (python bench.py, python bench.py --config5): the contract's rule"}),
              flush=True)
    ctx.close()


def config5_batch(a, ctx):
    """BASELINE config 5 as the timed workload (N > 1, or --config5): 512 stereo 10-min files at 192 kHz, one step =
    the whole archive.  Every rank pulls files (four per request, the last 4 x ranks of them one by one) from the shared longest-first queue
    (multi_gpu.WorkQueue: a host-side fetch-add, no collective) and runs them through resampling.varispeed_batch_dev: one
    plan per file (each file has its own speed curve: phase 0.7 + file index) and ONE stereo K_sinc launch,
    the next file's plan on a side stream under it.  Inputs resident in HBM before the timed region: a ring of `--ring`
    synthetic stereo files per GPU (the archive itself is 472 GB; signal content does not change the work) and the
    speed curves (7.2 MB per file, all of them on every rank: any rank may pull any file).

    Besides `value` (results left in HBM, like the N = 1 line) the line carries
    * the same workload on ONE GPU, measured in this run: rank 0 alone over `--n1-files` files of the archive while the
      other ranks wait -> n1_same_workload_value, speedup_vs_n1, efficiency;
    * the end-to-end leg (north_star: "results gathered on the host"; SURVEY 8e: kernel-only and end-to-end scaling
      reported separately): the same archive with every output copied into a pinned host ring per rank
      (resampling.varispeed_batch_gather) -> value_e2e, with its own one-GPU base."""
    import torch
    from pyaudiorestoration_amd import _dev, _lib, multi_gpu, resampling
    world, rank, dev = ctx.world, ctx.rank, ctx.local
    L = _lib.lib()
    s = _dev.stream_ptr(dev)
    numa = multi_gpu.bind_to_device_node(dev) if world > 1 else None      # pinned gather buffers on the GPU's own socket
    sr, seconds, files = a.sr, float(a.c5_seconds), a.files          # (--c5-seconds < 600: flow tests of many ranks on one device)
    n, m = int(sr * seconds), int(seconds * sr / 256)
    ring = []
    mono = torch.empty(n, dtype=torch.float32, device=f"cuda:{dev}")
    for k in range(a.ring):
        sig = torch.empty((n, 2), dtype=torch.float32, device=f"cuda:{dev}")
        for c in range(2):
            _lib.check(L.par_synth_signal_f32(dev, _dev.ptr(mono), 0, n, float(sr), 0x5EED ^ (2 * (rank * a.ring + k) + c), s))
            sig[:, c] = mono
        ring.append(sig)
    del mono
    # Speed curves: made per PULLED file (7.2 MB, a closed form on the device, ~2 us) into a small ring -- not all 512 of them
    # resident on every rank (3.7 GB each, r05).  The ring outlives the batch driver's prefetch (at most 16 items taken ahead
    # + a group of 8 in flight, see resampling.varispeed_batch_dev's PREFETCH CONTRACT): the kernel that fills slot k % 32 is
    # enqueued on the main stream behind the K_sinc of the file that last used it.
    n_curves = 32
    curves = torch.empty((n_curves, 2, m), dtype=torch.float64, device=f"cuda:{dev}")
    torch.cuda.synchronize()
    done = {"samples": 0, "files": 0}
    step_no = [0]

    def run_files(file_iter, gather):
        """One GPU's loop over the files it is handed -> (channel-samples, files)."""
        def produce():
            for k, f in enumerate(file_iter):
                c = curves[k % n_curves]
                _lib.check(L.par_synth_speed_curve_f64(dev, _dev.ptr(c[0]), _dev.ptr(c[1]), m, seconds, float(sr), 0.01, 0.55,
                                                       0.7 + f, _dev.stream_ptr(dev)))
                yield c[0], c[1], ring[k % a.ring]
        n_s = n_f = 0
        driver = resampling.varispeed_batch_gather if gather else resampling.varispeed_batch_dev
        for _, out, plan in driver(produce(), a.nt, dev):
            n_s += 2 * plan.len_out
            n_f += 1
        torch.cuda.synchronize(dev)
        return n_s, n_f

    def step(gather=False):
        q = multi_gpu.WorkQueue(ctx, range(files), f"s{step_no[0]}")
        step_no[0] += 1
        done["samples"], done["files"] = run_files(q, gather)

    def solo(n_files, gather):
        """Rank 0 alone over the first n_files files (the others idle at the barrier that follows): the one-GPU rate of
        THIS workload, taken in THIS run on THIS node."""
        rate = 0.0
        if rank == 0:
            # (an untimed pass of the same length first: this leg may follow seconds of set-up work, and a GPU timed cold reads
            # 5-15 % low for its first ~100 ms -- the base of the speed-up must not be the slow one)
            run_files(range(min(3, n_files) if gather else n_files), gather)
            t0 = time.perf_counter()
            n_s, _ = run_files(range(n_files), gather)
            rate = n_s / (time.perf_counter() - t0) / 1e6
        ctx.barrier()
        return rate

    # (rank 0's solo pass first: the other ranks idle meanwhile, and the warm-up steps should be what precedes the timed region)
    n1 = solo(min(files, a.n1_files), False)
    for _ in range(a.warmup):
        step()
    dt = ctx.timed(step, a.steps)
    total = ctx.reduce_sum(done["samples"])            # channel-samples of one step, all ranks
    files_max, files_min = ctx.reduce_max(done["files"]), -ctx.reduce_max(-done["files"])
    files_sum = ctx.reduce_sum(done["files"])          # every file of the archive exactly once: must equal `files`
    e2e = None
    if not a.no_e2e:
        n1_e2e = solo(min(files, a.n1_e2e_files), True)
        dt_e = ctx.timed(lambda: step(True), 1)
        total_e = ctx.reduce_sum(done["samples"])
        e2e = (total_e / dt_e / 1e6, n1_e2e, dt_e)
    # K_sinc of one file with this rank's GPU to itself (HIP events on the launch stream, 6 launches back to back behind 2 warm
    # ones): every rank measures its own device; the line carries the spread
    c0 = curves[0]
    _lib.check(L.par_synth_speed_curve_f64(dev, _dev.ptr(c0[0]), _dev.ptr(c0[1]), m, seconds, float(sr), 0.01, 0.55, 0.7 + rank, s))
    plan_k = resampling.speed_plan_dev(c0[0], c0[1], n, dev, fused=True)
    k_ms = 0.0
    if plan_k.fused_ok:
        item_k = (c0[0], c0[1], ring[0])
        for _ in range(2):
            resampling._resample_item(plan_k, item_k, a.nt, dev)      # (the batch driver's own launch: one stereo K_sinc)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(dev))
        for _ in range(6):
            resampling._resample_item(plan_k, item_k, a.nt, dev)
        ev1.record(torch.cuda.current_stream(dev))
        ev1.synchronize()
        k_ms = ev0.elapsed_time(ev1) / 6
    k_ms_max, k_ms_min = ctx.reduce_max(k_ms), -ctx.reduce_max(-k_ms)
    if rank == 0:
        value = total * a.steps / dt / 1e6
        res = {
            "metric": "Msamples/sec resampled (192 kHz varispeed)", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": ("f32 near taps, far taps (|n| >= 3) as float16 hi + lo x 2^-12 filter banks on v_mfma_f32_16x16x32_f16 with f32 accumulation / "
                                                                                      "f64 positions (interleaved stereo files at NT = 32: the streaming kernel's stereo form)" if a.nt == 32 else
                                                                                      "f32 taps / f64 positions (stereo files at this NT: the block kernel's vector tap loops)"),
            "data": "synthetic",
            "n1_same_workload_value": round(n1, 3), "speedup_vs_n1": round(value / n1, 3),
            "efficiency": round(value / n1 / world, 4), "distinct_devices": ctx.distinct_devices,
            "config": {"workload": f"config 5: {files}-file archive, {seconds:g}-s {sr} Hz stereo float32 each, +-1% sinusoidal speed "
                                   f"curve per file (0.55 Hz, hop 256, phase 0.7 + file index), {2 * a.nt}-tap Hann sinc; one step = the "
                                   "whole archive, files pulled by the ranks from a shared host-side queue (no collective, no RCCL)",
                       "files": files, "channel_samples_per_step": int(total), "files_per_rank_min_max": [int(files_min), int(files_max)],
                       "files_processed": int(files_sum),
                       "value_per_gpu": round(value / world, 3),
                       "n1_same_workload": f"n1_same_workload_value = rank 0 alone over {min(files, a.n1_files)} files of this archive, "
                                           "same code path, measured in this run before the timed region (the other ranks idle); "
                                           "speedup_vs_n1 = value / that, efficiency = speedup / n_gpus.  The `--gpus 1` line without "
                                           "--config5 times a different workload (the mono 60-min file of BASELINE configs[1]): do "
                                           "not build a curve from it",
                       "NT": a.nt, "resident": f"ring of {a.ring} synthetic stereo files per GPU; each file's speed curve is made when the file is pulled (ring of {n_curves})",
                       "queue": "one TCP-store fetch-add per 4 files; the last 4 x n_gpus files one per request (tail balance)",
                       "step": "per file: plan (device scans, block records; lazy: csrc/pos_plan.h) + ONE stereo fused K_sinc launch; "
                               "the plans of the next files are made by planner threads on side streams under K_sinc "
                               "(resampling.varispeed_batch_dev)"},
            "roofline": {"bound": "hbm", "achieved": round(ALGO_BYTES_PER_SAMPLE * value * 1e6 / 1e9 / world, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ALGO_BYTES_PER_SAMPLE * value * 1e6 / 1e9 / world / HBM_PEAK_GBS, 5),
                         "traffic": None, "kernel": "k_sinc stereo: k_sinc_pipe<2, 0> + k_sinc_fused_list2 at NT = 32 (whole step, per GPU)", "limited_by": "valu",
                         "kernel_ms_per_file_alone_min_max": [round(k_ms_min, 4), round(k_ms_max, 4)],
                         "kernel_frac_alone_min_max": ([round(ALGO_BYTES_PER_SAMPLE * 2 * plan_k.len_out / (k * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                                                        for k in (k_ms_max, k_ms_min)] if k_ms_min > 0 else None),
                         "note": "per-GPU whole-step rate x 8 algorithmic B per channel-sample against the HBM roof the contract "
                                 "names; what limits the kernel is VALU issue (the N = 1 line carries the per-kernel HIP-event "
                                 "timing, PMC traffic and the VALU roofline)"},
        }
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.sr, a.nt)          # (rank 0 at N = 1 only: the contract)
        else:
            res["cpu_baseline"] = None
            res["cpu_baseline_note"] = "timed on the N = 1 lines only (python bench.py, python bench.py --config5): the contract's rule"
        if e2e is not None:
            v_e, n1_e, dt_e = e2e
            res["value_e2e"] = round(v_e, 3)
            res["e2e"] = {"value_e2e": round(v_e, 3), "unit": "Msamples/s", "s_per_step": round(dt_e, 3), "steps": 1,
                          "n1_same_workload_value": round(n1_e, 3), "speedup_vs_n1": round(v_e / n1_e, 3),
                          "efficiency": round(v_e / n1_e / world, 4),
                          "GB_per_s_to_host_per_gpu": round(v_e * 1e6 * 4 / 1e9 / world, 2),
                          "rank0_numa_node": numa,
                          "what": "the same archive, every output copied to a ring of 3 pinned host buffers per rank on a "
                                  "separate stream under the next files' kernels (inputs stay resident in HBM); the bus (one "
                                  "0.92 GB D2H per file), not the kernels, sets this rate; one-GPU base = rank 0 alone over "
                                  f"{min(files, a.n1_e2e_files)} files"}
        print(json.dumps(res), flush=True)
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 200 mono files = 1.2 s of timed region at N = 1; 2 archive passes in the config-5 mode)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--seconds", type=float, default=3600.0, help="file duration (default: the 60-min config)")
    ap.add_argument("--sr", type=int, default=192000)
    ap.add_argument("--nt", type=int, default=32)
    ap.add_argument("--chunks", type=int, default=0, help="pipeline chunks per file (0 = auto, 1 = no overlap)")
    ap.add_argument("--no-fused", action="store_true", help="materialise the float64 position array (reference-shaped path)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="serial steps (plan, then K_sinc); default: the plan of file k+1 runs on a side stream under K_sinc of file k")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle windows over the timed output (A/B sessions)")
    ap.add_argument("--block-kernel", action="store_true", help="A/B: the block kernel instead of the streaming kernel (par_debug_sinc_kernel(0))")
    ap.add_argument("--config5", action="store_true", help="time the 512-file stereo archive (default when --gpus > 1)")
    ap.add_argument("--files", type=int, default=512, help="files of the config-5 archive")
    ap.add_argument("--c5-seconds", type=float, default=600.0, help="duration of the config-5 files (600 = the archive's; shorter: flow tests)")
    ap.add_argument("--ring", type=int, default=6, help="resident synthetic stereo files per GPU in the config-5 mode")
    ap.add_argument("--n1-files", type=int, default=64, help="files of the one-GPU same-workload base (config-5 mode)")
    ap.add_argument("--n1-e2e-files", type=int, default=24, help="files of the one-GPU base of the host-gather leg")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-gather leg of the config-5 mode")
    ap.add_argument("--dry-run", action="store_true", help="config-5 flow with a sleep per file instead of GPU work (CPU tests)")
    a = ap.parse_args()
    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if a.gpus > 1 and env_world == 0:
        sys.exit(_launch_ranks(a))                     # start the N ranks ourselves; rank 0 of the child run prints the line
    if env_world and a.gpus != env_world:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={env_world} ranks")
    batch_mode = a.gpus > 1 or a.config5 or a.dry_run
    if a.steps is None:
        a.steps = 2 if batch_mode else 200
    if a.warmup is None:
        a.warmup = 1 if batch_mode else 3
    if a.dry_run:
        from pyaudiorestoration_amd import multi_gpu
        return config5_dry_run(a, multi_gpu.RankContext())

    import torch
    from pyaudiorestoration_amd import _dev, _lib, multi_gpu

    ctx = multi_gpu.RankContext()                      # gloo (host-side barrier / reductions) when WORLD_SIZE > 1: no RCCL
    world, rank, local, dist = ctx.world, ctx.rank, ctx.local, ctx.dist
    if world > 1 or a.config5:
        return config5_batch(a, ctx)
    dev = local
    L = _lib.lib()
    sp_ = _dev.stream_ptr(dev)

    n_in = int(a.sr * a.seconds)
    m = int(a.seconds * a.sr / 256)
    sig = torch.empty(n_in, dtype=torch.float32, device=f"cuda:{dev}")
    st = torch.empty(m, dtype=torch.float64, device=f"cuda:{dev}")
    spd = torch.empty(m, dtype=torch.float64, device=f"cuda:{dev}")
    _lib.check(L.par_synth_signal_f32(dev, _dev.ptr(sig), 0, n_in, float(a.sr), 0x5EED ^ rank, sp_))
    _lib.check(L.par_synth_speed_curve_f64(dev, _dev.ptr(st), _dev.ptr(spd), m, a.seconds, float(a.sr), 0.01, 0.55,
                                           0.7 + rank, sp_))
    nbytes = int(L.par_speed_plan_bytes(m))
    cap = int(n_in * 1.02) + 1024
    out = torch.empty(cap, dtype=torch.float32, device=f"cuda:{dev}")
    fused = not a.no_fused
    overlap = fused and not a.no_overlap
    # Pipelined steps: K_sinc launches follow one another on the main stream; the plans of the next `depth` files are made by as
    # many planner threads, each on its own side stream (resampling.varispeed_batch_dev is the same driver for callers' items).
    # The streaming kernel leaves a plan no room beside it: plans advance in the gap behind a K_sinc -- `depth` of them together,
    # being latency-bound.  Depth 1 = the double-buffered pipeline of r02-r04; 3 measured best (ms per step at depth 1 / 2 / 3 / 4 / 8:
    # 5.0 / 4.66 / 4.57 / 4.63 / 4.63) and is the batch driver's default too.
    depth = max(1, int(os.environ.get("PAR_BENCH_DEPTH", "3"))) if overlap else 0
    n_slots = 2 * depth if overlap else 1
    work = [torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{dev}") for _ in range(n_slots)]
    if fused:       # cumsum checkpoints + tile map: positions are regenerated inside K_sinc, never stored
        aux_bytes = int(L.par_fused_aux_bytes(cap, m))
        aux = [torch.empty(aux_bytes, dtype=torch.uint8, device=f"cuda:{dev}") for _ in range(n_slots)]
    else:           # reference-shaped path: float64 sample_at array in HBM
        pos = torch.empty(cap, dtype=torch.float64, device=f"cuda:{dev}")
    torch.cuda.synchronize()

    sinc_ms, sinc_launches = [], []
    len_out = ctypes.c_int64(0)
    trimmed = ctypes.c_int(0)
    ok = ctypes.c_int(0)

    plan_force = 8 if os.environ.get("PAR_PLAN_EAGER") else 0      # A/B knob: the eager plan (per-sample cumsum + checkpoints)
    block_kernel = a.block_kernel or bool(os.environ.get("PAR_BENCH_BLOCK_KERNEL"))
    if block_kernel:
        L.par_debug_sinc_kernel(0)                                 # A/B knob: the block kernel for the mono NT = 32 file too
    streaming = fused and a.nt == 32 and not block_kernel          # what par_varispeed_fused_f32 launches for this workload
    kernel_symbols = (["k_sinc_pipe<1, 2>", "k_sinc_pipe<1, 1>", "k_sinc_fused_list"] if streaming else
                      [f"k_sinc_fused<1, {a.nt if a.nt in (32, 50) else 0}, 4>"] if fused else
                      [f"k_sinc_pos<{a.nt if a.nt in (32, 50) else 0}>"])
    state_plan = {"lazy": False}

    def plan_fused(slot, stream_ptr):
        _lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st), _dev.ptr(spd), m, n_in, _dev.ptr(work[slot]), nbytes,
                                                 _dev.ptr(aux[slot]), aux_bytes, cap, ctypes.byref(len_out),
                                                 ctypes.byref(trimmed), plan_force, None, ctypes.byref(ok), stream_ptr))
        assert ok.value in (1, 2) and 2 <= len_out.value <= cap      # 2: a lazy plan (csrc/pos_plan.h), what this curve gets
        state_plan["lazy"] = ok.value == 2
        return len_out.value

    if overlap:
        # Software pipeline over the files of a batch (one file per step): K_sinc of file k on the main stream,
        # the whole plan of file k+1 (25 small latency-bound kernels + one header read-back) on a side stream
        # underneath it.  Every step still executes one full plan and one full K_sinc; nothing is cached.
        import collections
        from concurrent.futures import ThreadPoolExecutor
        sides = [torch.cuda.Stream(device=dev, priority=int(os.environ.get("PAR_SIDE_PRIO", "0"))) for _ in range(depth)]
        slot_free = [None] * n_slots                    # main-stream event: K_sinc that read this slot is done
        pool = ThreadPoolExecutor(max_workers=depth)

        def plan_job(j):                                # the whole plan of file j (ctypes releases the GIL: planners run together)
            slot, stream = j % n_slots, sides[j % depth]
            if slot_free[slot] is not None:
                stream.wait_event(slot_free[slot])      # the K_sinc that last read this slot must be finished
            lo, tr, okv = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0)
            _lib.check(L.par_speed_to_pos_plan_fused(dev, _dev.ptr(st), _dev.ptr(spd), m, n_in, _dev.ptr(work[slot]), nbytes,
                                                     _dev.ptr(aux[slot]), aux_bytes, cap, ctypes.byref(lo), ctypes.byref(tr),
                                                     plan_force, None, ctypes.byref(okv), ctypes.c_void_p(stream.cuda_stream)))
            assert okv.value in (1, 2) and 2 <= lo.value <= cap
            state_plan["lazy"] = okv.value == 2
            return lo.value                             # (the call returned after its stream drained: header read-back)

        futs = collections.deque(pool.submit(plan_job, j) for j in range(depth))    # pipeline prologue: the first `depth` plans
        for f in futs:
            f.result()
        state = {"k": 0, "len": futs[0].result()}
        len_out.value = state["len"]
        ev_pairs = []
        # reference point outside the timed region: K_sinc with the GPU to itself (no plan underneath), launched back to
        # back like the timed steps -- 30 launches, the last 20 timed.  (A single launch on an idle GPU, which is what this
        # measured until r03, runs 8-10 % slower: the clocks have to come up first.  Sustained, the kernel sits at the
        # board's power limit, ~1.37 kW at 2.0-2.2 GHz: tools/exp/sustained_clock.py.)
        alone = []
        e0, e1, ms = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_float(0)
        _lib.check(L.par_event_create(ctypes.byref(e0)))
        _lib.check(L.par_event_create(ctypes.byref(e1)))
        for i in range(30):
            if i == 10:
                _lib.check(L.par_event_record(e0, sp_))
            _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(spd), m, _dev.ptr(work[0]), _dev.ptr(aux[0]), cap,
                                                 state["len"], _dev.ptr(sig), 1, n_in, a.nt, _dev.ptr(out), 1, sp_))
        _lib.check(L.par_event_record(e1, sp_))
        _lib.check(L.par_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
        alone.append(ms.value / 20)
        L.par_event_destroy(e0)
        L.par_event_destroy(e1)

        def step(timed):
            k = state["k"]
            cur = k % n_slots
            n_out = futs.popleft().result()             # the plan of file k
            e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
            if timed:
                _lib.check(L.par_event_create(ctypes.byref(e0)))
                _lib.check(L.par_event_create(ctypes.byref(e1)))
                _lib.check(L.par_event_record(e0, sp_))
            _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(spd), m, _dev.ptr(work[cur]), _dev.ptr(aux[cur]), cap,
                                                 n_out, _dev.ptr(sig), 1, n_in, a.nt, _dev.ptr(out), 1, sp_))
            if timed:
                _lib.check(L.par_event_record(e1, sp_))
                ev_pairs.append((e0, e1))
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            slot_free[cur] = ev
            futs.append(pool.submit(plan_job, k + depth))    # one plan per step: that of the file `depth` places on
            len_out.value = n_out
            state["k"] = k + 1
    else:
        _lib.check(L.par_profile_enable(dev, 1))         # HIP events around every K_sinc launch, on its own stream

        def step(timed):
            if fused:
                n_out = plan_fused(0, sp_)
                _lib.check(L.par_varispeed_fused_f32(dev, _dev.ptr(spd), m, _dev.ptr(work[0]), _dev.ptr(aux[0]), cap, n_out,
                                                     _dev.ptr(sig), 1, n_in, a.nt, _dev.ptr(out), 1, sp_))
            else:
                _lib.check(L.par_speed_to_pos_plan(dev, _dev.ptr(st), _dev.ptr(spd), m, n_in, _dev.ptr(work[0]), nbytes,
                                                   ctypes.byref(len_out), ctypes.byref(trimmed), sp_))
                assert 2 <= len_out.value <= cap
                _lib.check(L.par_varispeed_resample_f32(dev, _dev.ptr(spd), m, _dev.ptr(work[0]), len_out.value, _dev.ptr(pos),
                                                        _dev.ptr(sig), 1, n_in, a.nt, _dev.ptr(out), 1, a.chunks, sp_))
            if timed:
                ms, nl, ns = ctypes.c_float(0), ctypes.c_int(0), ctypes.c_int64(0)
                _lib.check(L.par_profile_read(dev, ctypes.byref(ms), ctypes.byref(nl), ctypes.byref(ns)))   # waits for them
                sinc_ms.append(ms.value)
                sinc_launches.append(nl.value)

    for _ in range(a.warmup):
        step(False)
    step_marks = []                                    # host clock at the end of every timed step (the steps end in a header
                                                       # read-back, i.e. a stream synchronisation: the marks are GPU-paced)

    def timed_step():
        step(True)
        step_marks.append(time.perf_counter())
    t_begin = time.perf_counter()
    dt = ctx.timed(timed_step, a.steps)                # barrier+sync | K steps | sync+barrier, MAX over ranks
    if overlap:                                        # (their GPU work ended inside the timed region's closing synchronise)
        for f in futs:
            f.result()
        pool.shutdown(wait=True)
    total_per_step = ctx.reduce_sum(len_out.value)     # whole-job output samples per step
    if overlap:                                        # K_sinc durations of the timed steps (events on its own stream)
        for e0, e1 in ev_pairs:
            ms = ctypes.c_float(0)
            _lib.check(L.par_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
            sinc_ms.append(ms.value)
            sinc_launches.append(1)
            L.par_event_destroy(e0)
            L.par_event_destroy(e1)

    if rank == 0:
        ms_step_mean = dt / a.steps * 1e3
        # per-step times from the marks (the first interval starts at the region's opening barrier); median of them
        marks = [t_begin] + step_marks
        per_step = sorted((b - a_) * 1e3 for a_, b in zip(marks[:-1], marks[1:]))
        if len(per_step) >= 3:
            per_step_mid = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
        else:
            per_step_mid = ms_step_mean
        # pipelined steps end when the host has QUEUED them, and plans become ready in bursts of `depth`: the marks are not
        # per-step GPU times any more and the region's mean is the step time; the serial modes keep the median of the marks
        ms_step = ms_step_mean if overlap else per_step_mid
        value = total_per_step * a.steps / dt / 1e6
        n_launch = sum(sinc_launches)
        k_ms = sum(sinc_ms) / n_launch                      # average K_sinc launch duration (HIP events)
        samples_per_launch = len_out.value * len(sinc_ms) / n_launch
        achieved = ALGO_BYTES_PER_SAMPLE * samples_per_launch / (k_ms * 1e-3) / 1e9
        # HBM traffic from the committed PMC passes (tools/profile_round.sh): bytes/sample x this run's rate
        traffic, traffic_src = None, None
        from pyaudiorestoration_amd import build as _build
        digest = _build.source_digest()                 # the kernel sources this run was built from
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            same = tr.get("kernel_symbols") == kernel_symbols      # the committed PMC pass measured the kernel(s) launched here
            if same:
                traffic = round(tr["hbm_bytes_per_sample"] * samples_per_launch / (k_ms * 1e-3) / 1e9, 2)
            traffic_src = {"file": "profiles/pmc_traffic.json", "from": tr.get("source"), "kernel_symbols": tr.get("kernel_symbols"),
                           "source_digest": tr.get("source_digest"), "running_digest": digest,
                           "stale": tr.get("source_digest") != digest or not same}
        except Exception:
            pass
        res = {
            "metric": "Msamples/sec resampled (192 kHz varispeed)", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 4),
            "ms_per_step_mean": round(ms_step_mean, 4),
            "ms_per_step_note": "ms_per_step = timed region / steps, the figure `value` is computed from (pipelined steps are queued "
                                "ahead of the GPU, so host marks per step say nothing; --no-overlap reports the median of its per-step times)",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 near taps, far taps (|n| >= 3) as float16 hi + lo x 2^-12 filter banks on v_mfma_f32_16x16x32_f16 with f32 "
                      "accumulation / f64 positions" if streaming else
                      "f32 taps (fc = 1 far taps f16 hi + lo on MFMA) / f64 positions" if fused and a.nt == 32 else "f32 taps / f64 positions"),
            "data": "synthetic",
            "config": {"workload": f"{a.seconds:g}-s {a.sr} Hz mono float32 varispeed resample, +-1% sinusoidal speed "
                                   f"curve (0.55 Hz, hop 256), {2 * a.nt}-tap Hann sinc; one file per GPU",
                       "samples_in_per_gpu": n_in, "samples_out_per_gpu": int(len_out.value), "NT": a.nt,
                       "plan": ("lazy (closed-form segment sums, exact ones for the offset chain's candidates: csrc/pos_plan.h)" if state_plan["lazy"] else "eager (per-sample cumsum + checkpoints)") if fused else "position array",
                       "step": ("plan (device scans, block records) + fused K_sinc (outputs placed from 16-byte block records, no position array)" if fused else "plan (device scans) + K_pos fill (float64 position array) + K_sinc") + "; inputs resident in HBM"
                               + (f"; batch pipelining: K_sinc launches back to back, the plans of the next {depth} files by {depth} planner threads on side streams (every step = one full plan + one full K_sinc)" if overlap else ""),
                       "planners": depth},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "source": traffic_src,
                         "kernel": "k_sinc", "kernel_symbols": kernel_symbols, "limited_by": "valu",
                         "warm_launches_before_timed": (30 if overlap else 0) + a.warmup,
                         "kernel_ms": round(k_ms, 4), "launches_per_step": n_launch // len(sinc_ms),
                         "samples_per_launch": int(samples_per_launch),
                         "note": "achieved = 8 algorithmic B/output sample (4 B in + 4 B out) / HIP-event K_sinc time (events on the launch stream "
                                 "around the whole launch: for the streaming path its two or three kernels AND the dispatch gaps between them, "
                                 "~0.1-0.2 ms beside the planners' kernels -- the per-kernel durations of profiles/r06_kernel_stats.txt sum to "
                                 "that much less); "
                                 "traffic = PMC HBM bytes/sample (FETCH_SIZE x2 + WRITE_SIZE, profiles/pmc_traffic.json: a separate "
                                 "rocprofv3 --pmc pass, stamped with the digest of the kernel sources it ran; `source.stale` says "
                                 "whether that is the code running now) at this run's rate" + (": signal + output + 1 B/sample cumsum checkpoints + tile halos"
                                                 if fused else " incl. the 8 B float64 position read") +
                                 "; `bound` names the roof this object measures against (the contract's HBM roof on algorithmic bytes); the kernel is LIMITED BY VALU issue (64 taps per output against 8 B): roofline_valu is the roof it sits under; FETCH_SIZE x2 calibrated for this kernel's access patterns in profiles/r02_fetch_calibration.txt"},
        }
        if overlap:
            k_alone = min(alone)
            res["roofline"]["kernel_ms_alone"] = round(k_alone, 4)
            res["roofline"]["frac_alone"] = round(ALGO_BYTES_PER_SAMPLE * samples_per_launch / (k_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            res["roofline"]["note"] += ("; kernel_ms is measured in the timed region, where the next file's plan kernels "
                                        "share the GPU with K_sinc; kernel_ms_alone / frac_alone = the same launch with the "
                                        "GPU to itself, 20 launches back to back before the timed region")
        # What actually limits the kernel: VALU issue.  Instruction count per output from the committed PMC pass
        # (SQ_INSTS_VALU, profiles/), rate from this run's HIP-event time; ceilings: 2 cycles per wave64 instruction
        # per SIMD at the 2.4 GHz peak clock (MI355X_MICROARCH.md) and the rate a pure v_fma_f32 stream measured
        # (tools/ubench.hip, profiles/r01_ubench_gfx950.txt).
        try:
            vp = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu.json")))
            if vp.get("kernel_symbols") != kernel_symbols:          # counters of some other kernel: say nothing rather than that
                raise KeyError("pmc_valu.json describes " + str(vp.get("kernel_symbols")))
            ipo = vp["valu_lane_instr_per_output"]
            ach = ipo * samples_per_launch / (k_ms * 1e-3) / 1e12
            res["roofline_valu"] = {"limited_by": "valu", "valu_lane_instr_per_output": round(ipo, 1),
                                    "achieved_Tlaneops": round(ach, 2), "ceiling_spec_Tlaneops": 78.64,
                                    "ceiling_measured_fma_stream_Tlaneops": vp.get("fma_stream_Tlaneops", 58.76),
                                    "frac_of_spec": round(ach / 78.64, 4),
                                    "frac_of_measured": round(ach / vp.get("fma_stream_Tlaneops", 58.76), 4),
                                    "kernel_symbols": kernel_symbols,
                                    "source": vp.get("source", "profiles/"), "source_digest": vp.get("source_digest"),
                                    "running_digest": digest, "stale": vp.get("source_digest") != digest}
        except Exception:
            pass
        if world == 1:
            res["secondary"] = stft_secondary(sig, dev, cpu=not a.no_cpu_baseline)
            res["secondary_config4"] = heal_secondary(dev)
            if hasattr(L, "par_varispeed_fused_stereo_f32"):        # absent only in an older build under PAR_HIP_LIB
                res["secondary_config5"] = stereo_secondary(dev)
            if fused and a.nt == 32:
                # (a fresh plan of the timed curve on the main stream: the pipelined steps' plan slots have moved on)
                n50 = plan_fused(0, sp_)
                res["secondary_nt50"] = nt50_secondary(dev, sig, st, spd, n_in, work[0], aux[0], cap, n50)
                # the 1-GPU point of the --gpus N > 1 curve under a stable key: N > 1 lines time the config-5 archive (stereo
                # files through the batch driver), not this line's mono file -- a scaling reader divides value(N) by THIS
                res["archive_value"] = res["secondary_config5"]["batched_Msamples/s"]
                res["archive_unit"] = "M channel-samples/s"
                res["archive_ms_per_file"] = res["secondary_config5"]["batched_ms_per_file"]
                res["archive_note"] = ("config-5 archive rate of ONE GPU (10-min 192 kHz stereo files through the batch driver, the "
                                       "code path of --gpus N > 1): the base of the 1/2/4/8-GPU curve is archive_value here and "
                                       "n1_same_workload_value in the N > 1 lines, never this line's `value`")
        if not a.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(a.sr, a.nt)
        if not a.no_parity and world == 1:
            torch.cuda.synchronize()
            res["parity_checked"] = parity_windows(out, sig, st, spd, n_in, int(len_out.value), a.nt)
        print(json.dumps(res), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
