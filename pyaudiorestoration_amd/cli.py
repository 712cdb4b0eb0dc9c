"""Headless pyrespeeder: a fresh replacement for the reference's stale
experiments/pyrespeeder_cmd.py.  Files are independent work items: one host thread per GPU pulls
(file) items from a queue, every file goes STFT -> tracker -> master speed curve -> fused sinc
resample on that GPU, results are written as <stem>_res<suffix>.wav like resampling.run does
(util/resampling.py:235-237).  No Qt, no collective communication.

    python -m pyaudiorestoration_amd.cli respeed --trail 0.2,4000,4.0,4000 tape1.flac tape2.wav
    python -m pyaudiorestoration_amd.cli respeed --project tape.spd tape.flac      # traces / regressions saved by the GUI
    python -m pyaudiorestoration_amd.cli resample --curve curve.json tape.wav      # [[t_seconds, speed], ...]
    python -m pyaudiorestoration_amd.cli resample --speed 1.015 tape.wav           # constant correction
    python -m pyaudiorestoration_amd.cli tapesync --project take.tapesync take2.flac
    python -m pyaudiorestoration_amd.cli heal --project tape.drop tape.flac
"""
import argparse
import json
import logging
import os
import queue
import sys
import threading

import numpy as np


def _worker(dev, jobs, args, results):
    """One host thread per GPU.  The next file of the queue is read and decoded (native FLAC decoder / numpy, both
    release the GIL) on a helper thread while the current one is on the GPU."""
    from concurrent.futures import ThreadPoolExecutor
    import torch
    from . import _dev, io_ops, pipeline, resampling
    torch.cuda.set_device(dev)

    def take():
        try:
            path = jobs.get_nowait()
        except queue.Empty:
            return None
        if args.cmd in ("tapesync", "heal") or (args.cmd == "respeed" and args.project):        # these flows read their source themselves
            return path, None
        return path, reader.submit(io_ops.read_file, path)

    with ThreadPoolExecutor(max_workers=1) as reader:
        nxt = take()
        while nxt is not None:
            (path, pending), nxt = nxt, take()       # the next file starts decoding now
            try:
                if args.cmd in ("tapesync", "heal") or (args.cmd == "respeed" and args.project):
                    flow = {"tapesync": pipeline.tapesync, "heal": pipeline.heal_project, "respeed": pipeline.respeed_project}[args.cmd]
                    if args.cmd == "respeed":       # --suffix / --quality / --resampling override the project's own only when given
                        flow(args.project, source=path, out_suffix=args.suffix, device=dev, sinc_quality=args.quality,
                             resampling_mode=getattr(args, "resampling", None))
                    else:
                        flow(args.project, source=path, out_suffix=args.suffix, device=dev)
                    results.append((path, None))
                    continue
                signal, sr, ch = pending.result()
                quality = 50 if args.quality is None else args.quality          # the GUI's default (util/widgets.py:998-1000)
                suffix = args.suffix or ""
                if args.cmd == "respeed":
                    t0, f0, t1, f1 = args.trail
                    r = pipeline.respeed(signal, sr, [(t0, f0), (t1, f1)], args.fft_size, args.hop, 1, args.mode,
                                         args.tolerance, (0, args.lowpass), quality, device=dev)
                    out = _dev.to_host(r["output"])
                    io_ops.write_wav_float(f"{os.path.splitext(path)[0]}_res{suffix}.wav", out, sr)
                    np.save(f"{os.path.splitext(path)[0]}_speed{suffix}.npy", r["speed_curve"])
                else:
                    if args.speed is not None:          # constant correction: a two-point curve over the whole file
                        curve = np.array([[0.0, args.speed], [len(signal) / sr, args.speed]], dtype=np.float64)
                    else:
                        curve = np.asarray(json.load(open(args.curve)), dtype=np.float64)
                    resampling.run((path,), signal_data=((signal, sr),), speed_curve=curve, resampling_mode=args.resampling,
                                   sinc_quality=quality, suffix=suffix)
                results.append((path, None))
            except Exception as e:                      # keep the other files going; report at the end
                logging.exception(f"{path} failed")
                results.append((path, e))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="pyaudiorestoration_amd.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("respeed", help="trace a pilot tone / hum and remove wow & flutter")
    what = a.add_mutually_exclusive_group(required=True)
    what.add_argument("--trail", type=lambda s: [float(v) for v in s.split(",")], help="t0,f0,t1,f1 (s, Hz): trace it here")
    what.add_argument("--project", help=".spd JSON written by the GUI: its traces and regressions make the master curve")
    a.add_argument("--mode", default="Peak", help="tracker name as in wow_detection.wow_detectors")
    a.add_argument("--tolerance", type=float, default=0.5, help="semitones")
    a.add_argument("--fft-size", type=int, default=1024)
    a.add_argument("--hop", type=int, default=256)
    a.add_argument("--lowpass", type=float, default=20.0, help="speed-curve low-pass (Hz)")
    b = sub.add_parser("resample", help="apply a given speed curve")
    how = b.add_mutually_exclusive_group(required=True)
    how.add_argument("--curve", help="JSON [[t_seconds, speed], ...]")
    how.add_argument("--speed", type=float, help="constant speed factor of the recording (1.015 = it ran 1.5 %% fast)")
    b.add_argument("--resampling", default="Sinc", choices=("Sinc", "Linear"))
    c = sub.add_parser("tapesync", help="apply the lag curve of a saved pytapesynch project (.tapesync) to files")
    c.add_argument("--project", required=True, help=".tapesync JSON written by the GUI")
    d = sub.add_parser("heal", help="inpaint the marked dropouts of a saved dropout-healer project (.drop)")
    d.add_argument("--project", required=True, help=".drop JSON written by the GUI")
    for p in (a, b, c, d):
        p.add_argument("--quality", type=int, default=None, help="sinc_quality (NT); default: the project's, else the GUI's 50")
        p.add_argument("--suffix", default=None, help="output suffix; default: the project's, else none")
        p.add_argument("--gpus", type=int, default=0, help="GPUs to use (0 = all visible)")
        p.add_argument("files", nargs="+")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(levelname)s %(message)s")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("no ROCm GPU visible; this tool has no CPU fallback")
    n_gpu = torch.cuda.device_count() if args.gpus <= 0 else min(args.gpus, torch.cuda.device_count())
    jobs = queue.Queue()
    for f in sorted(args.files, key=lambda p: -os.path.getsize(p)):     # longest first
        jobs.put(f)
    results = []
    threads = [threading.Thread(target=_worker, args=(d, jobs, args, results)) for d in range(n_gpu)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    failed = [p for p, e in results if e is not None]
    logging.info(f"{len(results) - len(failed)} file(s) done on {n_gpu} GPU(s), {len(failed)} failed")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
