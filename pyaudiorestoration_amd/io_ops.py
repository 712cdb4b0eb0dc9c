"""Minimal audio file I/O standing in for soundfile/libsndfile (reference util/io_ops.py:7-23,
util/resampling.py:235-237): RIFF/WAVE reader (PCM 16/24/32, IEEE float 32/64) returning float32
``(frames, channels)`` like ``SoundFile.read(always_2d=True, dtype="float32")``, a self-checking FLAC decoder for the reference's
sample files, and an IEEE-float WAV writer (libsndfile subtype 'FLOAT').  Host-side plumbing only -- no DSP."""
import logging
import os
import struct

import numpy as np


def read_wav(path):
    """RIFF/WAVE -> (float32 (frames, channels), sr, channels).  The chunk headers are walked with seeks and the data chunk is read
    ONCE, straight into the array (r06: the whole file as bytes, a slice of it and a converted copy were three passes over 460 MB
    for a 10-min file)."""
    with open(path, "rb") as f:
        head = f.read(12)
        if head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise ValueError(f"{path}: not a RIFF/WAVE file")
        size_of_file = os.fstat(f.fileno()).st_size
        fmt = None
        data_at = data_size = None
        pos = 12
        while pos + 8 <= size_of_file:
            f.seek(pos)
            hdr = f.read(8)
            if len(hdr) < 8:
                break
            cid, size = hdr[:4], struct.unpack("<I", hdr[4:8])[0]
            if cid == b"fmt ":
                body = f.read(min(size, 64))
                tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
                if tag == 0xFFFE and len(body) >= 26:          # WAVE_FORMAT_EXTENSIBLE: real tag in the GUID
                    tag = struct.unpack("<H", body[24:26])[0]
                fmt = (tag, ch, sr, bits)
            elif cid == b"data":
                data_at, data_size = pos + 8, min(size, size_of_file - pos - 8)      # (a truncated file: what is there)
            pos += 8 + size + (size & 1)
        if fmt is None or data_at is None:
            raise ValueError(f"{path}: missing fmt/data chunk")
        tag, ch, sr, bits = fmt
        kinds = {(3, 32): "<f4", (3, 64): "<f8", (1, 16): "<i2", (1, 32): "<i4", (1, 24): np.uint8}
        if (tag, bits) not in kinds:
            raise ValueError(f"{path}: unsupported WAV format tag={tag} bits={bits}")
        dt = np.dtype(kinds[(tag, bits)])
        f.seek(data_at)
        raw = np.fromfile(f, dtype=dt, count=data_size // dt.itemsize)
    if tag == 3 and bits == 32:
        x = raw.astype(np.float32, copy=False)
    elif tag == 3 and bits == 64:
        x = raw.astype(np.float32)
    elif tag == 1 and bits == 16:
        x = raw.astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = (raw.astype(np.float64) / 2147483648.0).astype(np.float32)
    else:                                                      # 24-bit PCM
        b = raw[:len(raw) // 3 * 3].reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        x = (v / 8388608.0).astype(np.float32)
    frames = len(x) // ch
    return x[:frames * ch].reshape(frames, ch), sr, ch


def write_wav_float(path, signal, sr):
    """IEEE float32 WAV (what sf.SoundFile(..., subtype='FLOAT') writes): the header, then the samples from the array's own
    memory (r06: no bytes copies of the payload)."""
    signal = np.asarray(signal, dtype=np.float32)
    if signal.ndim == 1:
        signal = signal[:, None]
    frames, ch = signal.shape
    payload = np.ascontiguousarray(signal, dtype="<f4")
    nbytes = payload.nbytes
    if nbytes + 50 >= 1 << 32:
        raise ValueError(f"{path}: {nbytes} bytes of samples do not fit a RIFF/WAVE file (4 GiB)")
    fmt = struct.pack("<HHIIHH", 3, ch, int(sr), int(sr) * ch * 4, ch * 4, 32)
    fact = struct.pack("<I", frames)
    head = (b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"fact" + struct.pack("<I", 4) + fact
            + b"data" + struct.pack("<I", nbytes))
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(head) + nbytes) + head)
        payload.tofile(f)


def read_file(audio_path):
    logging.info(f"Reading {audio_path}")
    ext = os.path.splitext(audio_path)[1].lower()
    if ext == ".flac":
        signal, sr, channels = read_flac(audio_path)
    elif ext == ".wav":
        signal, sr, channels = read_wav(audio_path)
    else:
        raise NotImplementedError(f"{audio_path}: only WAV and FLAC are decoded here (no libsndfile in this image)")
    if len(signal) == 0:
        raise AttributeError(f"Reading {audio_path} failed")
    return signal, sr, channels


def write_file(audio_path, signal, sr, channels, suffix="_out"):
    write_wav_float(f"{os.path.splitext(audio_path)[0]}{suffix}.wav", signal, sr)
    logging.info(f"Wrote {audio_path}")


# ------------------------------------------------------------------------------------------- FLAC
# Minimal pure-Python FLAC decoder (subset the format defines for lossless PCM: CONSTANT / VERBATIM /
# FIXED / LPC subframes, Rice-coded residuals incl. escape partitions, independent and left/right/mid-side
# stereo).  Stands in for libsndfile so that read_file() can open the reference's sample files
# (util/io_ops.py:7-16).  Self-checking: the decoded PCM is verified against the MD5 in STREAMINFO.

class _BitReader:
    def __init__(self, data, pos=0):
        self.data = data
        self.pos = pos * 8                   # bit position

    def read(self, n):
        if n == 0:
            return 0
        p = self.pos
        first, last = p >> 3, (p + n + 7) >> 3
        v = int.from_bytes(self.data[first:last], "big")
        v >>= (last << 3) - (p + n)
        self.pos = p + n
        return v & ((1 << n) - 1)

    def read_signed(self, n):
        v = self.read(n)
        return v - (1 << n) if n and v >> (n - 1) else v

    def unary(self):
        """number of 0 bits before the next 1 bit (consumes the 1)."""
        count = 0
        while True:
            p = self.pos
            byte_i, bit = p >> 3, p & 7
            chunk = int.from_bytes(self.data[byte_i:byte_i + 8], "big")
            nb = min(8, len(self.data) - byte_i) * 8
            if nb == 0:
                raise ValueError("FLAC: ran off the end of the stream")
            chunk &= (1 << (nb - bit)) - 1
            if chunk:
                lz = (nb - bit) - chunk.bit_length()
                self.pos = p + lz + 1
                return count + lz
            count += nb - bit
            self.pos = p + nb - bit

    def align(self):
        self.pos = (self.pos + 7) & ~7


def _flac_residual(br, blocksize, order, out):
    method = br.read(2)
    if method > 1:
        raise ValueError("FLAC: reserved residual coding method")
    pbits = 4 if method == 0 else 5
    porder = br.read(4)
    nparts = 1 << porder
    for part in range(nparts):
        cnt = (blocksize >> porder) - (order if part == 0 else 0)
        k = br.read(pbits)
        if k == (1 << pbits) - 1:                       # escape: raw signed samples
            nb = br.read(5)
            for _ in range(cnt):
                out.append(br.read_signed(nb))
        else:
            unary, read = br.unary, br.read
            for _ in range(cnt):
                u = (unary() << k) | read(k)
                out.append((u >> 1) ^ -(u & 1))


_FIXED = {0: (), 1: (1,), 2: (2, -1), 3: (3, -3, 1), 4: (4, -6, 4, -1)}


def _flac_subframe(br, blocksize, bps):
    if br.read(1):
        raise ValueError("FLAC: subframe padding bit set")
    typ = br.read(6)
    wasted = 0
    if br.read(1):
        wasted = br.unary() + 1
        bps -= wasted
    if typ == 0:
        s = [br.read_signed(bps)] * blocksize
    elif typ == 1:
        s = [br.read_signed(bps) for _ in range(blocksize)]
    elif 8 <= typ <= 12:
        order = typ - 8
        s = [br.read_signed(bps) for _ in range(order)]
        res = []
        _flac_residual(br, blocksize, order, res)
        co = _FIXED[order]
        for r in res:
            s.append(r + sum(c * s[-1 - i] for i, c in enumerate(co)))
    elif typ >= 32:
        order = (typ & 31) + 1
        s = [br.read_signed(bps) for _ in range(order)]
        prec = br.read(4) + 1
        shift = br.read_signed(5)
        co = [br.read_signed(prec) for _ in range(order)]
        res = []
        _flac_residual(br, blocksize, order, res)
        for r in res:
            acc = 0
            for i, c in enumerate(co):
                acc += c * s[-1 - i]
            s.append(r + (acc >> shift))
    else:
        raise ValueError(f"FLAC: reserved subframe type {typ}")
    if wasted:
        s = [v << wasted for v in s]
    return s


def read_flac(path, verify_md5=True, n_threads=0):
    """Decode a FLAC file -> (float32 (frames, channels) scaled like libsndfile, samplerate, channels) with the
    library's native frame-parallel decoder (csrc/flac_host.hip; host code, no GPU involved)."""
    import ctypes
    from . import _lib
    L = _lib.lib()
    data = np.fromfile(path, dtype=np.uint8)
    buf = data.ctypes.data_as(ctypes.c_void_p)
    sr, ch, bits, total = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int64(0)
    _lib.check(L.par_flac_info(buf, data.size, ctypes.byref(sr), ctypes.byref(ch), ctypes.byref(bits), ctypes.byref(total), None))
    # a frame is at least 10 bytes (header 6, one constant subframe 2, CRC-16 2) and holds at most 65 535 samples per
    # channel: a STREAMINFO that claims more than the file can hold is corrupt -- refuse it before allocating for it
    if total.value > (data.size // 8 + 1) * 65535:
        raise ValueError(f"{path}: STREAMINFO claims {total.value} samples, more than a {data.size}-byte file can hold")
    out = np.empty((total.value, ch.value), dtype=np.float32)
    done = ctypes.c_int64(0)
    _lib.check(L.par_flac_decode_f32(buf, data.size, out.ctypes.data_as(ctypes.c_void_p), total.value, int(n_threads),
                                     1 if verify_md5 else 0, ctypes.byref(done)))
    return out, sr.value, ch.value


def read_flac_py(path, verify_md5=True):
    """The same decoder in pure Python (~1 Msample/s): an independent implementation the tests hold the native
    one against."""
    import hashlib
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"fLaC":
        raise ValueError(f"{path}: not a FLAC stream")
    pos = 4
    info = None
    while True:
        hdr = data[pos]
        length = int.from_bytes(data[pos + 1:pos + 4], "big")
        body = data[pos + 4:pos + 4 + length]
        if hdr & 0x7F == 0:
            v = int.from_bytes(body[10:18], "big")
            info = {"sr": v >> 44, "ch": ((v >> 41) & 7) + 1, "bps": ((v >> 36) & 31) + 1, "total": v & ((1 << 36) - 1),
                    "md5": body[18:34], "maxblock": int.from_bytes(body[2:4], "big")}
        pos += 4 + length
        if hdr & 0x80:
            break
    if info is None:
        raise ValueError(f"{path}: no STREAMINFO")
    ch, bps, total = info["ch"], info["bps"], info["total"]
    chans = [[] for _ in range(ch)]
    br = _BitReader(data, pos)
    nbytes = len(data)
    while (br.pos >> 3) < nbytes and (total == 0 or len(chans[0]) < total):
        if br.read(14) != 0x3FFE:
            raise ValueError("FLAC: lost frame sync")
        br.read(1)
        br.read(1)                                      # blocking strategy (only matters for the number below)
        bs_code, sr_code = br.read(4), br.read(4)
        ch_code, ss_code = br.read(4), br.read(3)
        br.read(1)
        lead = br.read(8)                               # UTF-8 style frame / sample number
        nfollow = 0
        while lead & (0x80 >> nfollow):
            nfollow += 1
        for _ in range(max(0, nfollow - 1)):
            br.read(8)
        if bs_code == 1:
            blocksize = 192
        elif 2 <= bs_code <= 5:
            blocksize = 576 << (bs_code - 2)
        elif bs_code == 6:
            blocksize = br.read(8) + 1
        elif bs_code == 7:
            blocksize = br.read(16) + 1
        elif bs_code >= 8:
            blocksize = 256 << (bs_code - 8)
        else:
            raise ValueError("FLAC: reserved block size")
        if sr_code == 12:
            br.read(8)
        elif sr_code in (13, 14):
            br.read(16)
        br.read(8)                                      # CRC-8
        fbps = {0: bps, 1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}.get(ss_code)
        if fbps is None:
            raise ValueError("FLAC: reserved sample size")
        if ch_code < 8:
            subs = [_flac_subframe(br, blocksize, fbps) for _ in range(ch_code + 1)]
        elif ch_code == 8:                              # left / side
            left, side = _flac_subframe(br, blocksize, fbps), _flac_subframe(br, blocksize, fbps + 1)
            subs = [left, [a - b for a, b in zip(left, side)]]
        elif ch_code == 9:                              # side / right
            side, right = _flac_subframe(br, blocksize, fbps + 1), _flac_subframe(br, blocksize, fbps)
            subs = [[a + b for a, b in zip(side, right)], right]
        elif ch_code == 10:                             # mid / side
            mid, side = _flac_subframe(br, blocksize, fbps), _flac_subframe(br, blocksize, fbps + 1)
            left = [(((m << 1) | (s & 1)) + s) >> 1 for m, s in zip(mid, side)]
            subs = [left, [(((m << 1) | (s & 1)) - s) >> 1 for m, s in zip(mid, side)]]
        else:
            raise ValueError("FLAC: reserved channel assignment")
        br.align()
        br.read(16)                                     # CRC-16
        for c in range(ch):
            chans[c].extend(subs[c])
    pcm = np.array(chans, dtype=np.int64).T            # (frames, channels)
    if total:
        pcm = pcm[:total]
    if verify_md5 and any(info["md5"]):
        width = (bps + 7) // 8
        raw = pcm.astype("<i8").reshape(-1, 1).view(np.uint8)[:, :width].tobytes()
        if hashlib.md5(raw).digest() != info["md5"]:
            raise ValueError(f"{path}: decoded PCM does not match the STREAMINFO MD5")
    return (pcm / float(1 << (bps - 1))).astype(np.float32), info["sr"], ch
