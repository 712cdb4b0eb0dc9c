"""Minimal audio file I/O standing in for soundfile/libsndfile (reference util/io_ops.py:7-23,
util/resampling.py:235-237): RIFF/WAVE reader (PCM 16/24/32, IEEE float 32/64) returning float32
``(frames, channels)`` like ``SoundFile.read(always_2d=True, dtype="float32")``, and an IEEE-float
WAV writer (libsndfile subtype 'FLOAT').  Host-side plumbing only -- no DSP here."""
import logging
import os
import struct

import numpy as np


def read_wav(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos = 12
    fmt = None
    raw = None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:          # WAVE_FORMAT_EXTENSIBLE: real tag in the GUID
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            raw = body
        pos += 8 + size + (size & 1)
    if fmt is None or raw is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, bits = fmt
    if tag == 3 and bits == 32:
        x = np.frombuffer(raw, dtype="<f4").astype(np.float32)
    elif tag == 3 and bits == 64:
        x = np.frombuffer(raw, dtype="<f8").astype(np.float32)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif tag == 1 and bits == 24:
        b = np.frombuffer(raw[:len(raw) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        x = (v / 8388608.0).astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAV format tag={tag} bits={bits}")
    frames = len(x) // ch
    return x[:frames * ch].reshape(frames, ch), sr, ch


def write_wav_float(path, signal, sr):
    """IEEE float32 WAV (what sf.SoundFile(..., subtype='FLOAT') writes)."""
    signal = np.asarray(signal, dtype=np.float32)
    if signal.ndim == 1:
        signal = signal[:, None]
    frames, ch = signal.shape
    payload = np.ascontiguousarray(signal, dtype="<f4").tobytes()
    fmt = struct.pack("<HHIIHH", 3, ch, int(sr), int(sr) * ch * 4, ch * 4, 32)
    fact = struct.pack("<I", frames)
    body = (b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"fact" + struct.pack("<I", 4) + fact
            + b"data" + struct.pack("<I", len(payload)) + payload)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def read_file(audio_path):
    logging.info(f"Reading {audio_path}")
    ext = os.path.splitext(audio_path)[1].lower()
    if ext != ".wav":
        raise NotImplementedError(f"{audio_path}: only WAV is decoded here (no libsndfile in this image)")
    signal, sr, channels = read_wav(audio_path)
    if len(signal) == 0:
        raise AttributeError(f"Reading {audio_path} failed")
    return signal, sr, channels


def write_file(audio_path, signal, sr, channels, suffix="_out"):
    write_wav_float(f"{os.path.splitext(audio_path)[0]}{suffix}.wav", signal, sr)
    logging.info(f"Wrote {audio_path}")
