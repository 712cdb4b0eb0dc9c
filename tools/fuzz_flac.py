#!/usr/bin/env python3
"""Robustness fuzz of the native FLAC decoder (host code, runs anywhere): valid streams from the test-side encoder with
random byte flips, truncations, garbage tails and bogus STREAMINFO fields.  The decoder must either return the exact
PCM or raise ParError -- never crash, hang, or return wrong samples silently (the MD5 check is on)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tempfile

import numpy as np

import flac_writer
from pyaudiorestoration_amd import _lib, io_ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t_end = time.time() + budget
case = ok = refused = 0
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "f.flac")
while time.time() < t_end:
    rng = np.random.default_rng(case)
    ch = int(rng.choice([1, 2]))
    bps = int(rng.choice([8, 16, 24]))
    frames = int(rng.choice([300, 5000, 40000]))
    pcm = np.rint(rng.normal(0, (1 << (bps - 1)) * 0.1, (frames, ch))).astype(np.int64)
    pcm = np.clip(pcm, -(1 << (bps - 1)), (1 << (bps - 1)) - 1)
    blob = bytearray(flac_writer.encode_flac(pcm, 44100, bps, blocksize=int(rng.choice([256, 1024, 4096])),
                                             stereo="mid_side" if ch == 2 and rng.random() < 0.5 else "independent",
                                             kind=str(rng.choice(["fixed", "lpc", "verbatim"])), order=int(rng.integers(1, 5)),
                                             variable=bool(rng.random() < 0.3)))
    mode = int(rng.integers(0, 6))
    if mode == 0:
        pass                                                        # untouched: must decode exactly
    elif mode == 1:
        for _ in range(int(rng.integers(1, 6))):
            blob[int(rng.integers(0, len(blob)))] ^= 1 << int(rng.integers(0, 8))
    elif mode == 2:
        blob = blob[:int(rng.integers(4, len(blob)))]
    elif mode == 3:
        blob += bytes(rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8))
    elif mode == 4:
        s = int(rng.integers(42, len(blob) - 1))
        blob[s:s + int(rng.integers(1, 400))] = bytes(rng.integers(0, 256, 1, dtype=np.uint8)) * 1
    else:
        blob[8 + int(rng.integers(0, 34))] = int(rng.integers(0, 256))   # STREAMINFO field
    open(path, "wb").write(bytes(blob))
    want = (pcm / float(1 << (bps - 1))).astype(np.float32)
    try:
        got, sr, n_ch = io_ops.read_flac(path, n_threads=int(rng.choice([1, 4])))
    except (_lib.ParError, MemoryError, ValueError):
        assert mode != 0, (case, "a valid stream was refused")
        refused += 1
    else:
        # accepted: with the MD5 check on, the samples must be the original ones (a garbage tail after the last frame
        # and flips inside the padding block are harmless)
        assert got.shape == want.shape and np.array_equal(got, want), (case, mode, "accepted a corrupted stream with wrong samples")
        ok += 1
    case += 1
print(f"flac fuzz ok: {case} streams, {ok} decoded exactly, {refused} refused cleanly")
