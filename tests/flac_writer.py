"""Test-side FLAC ENCODER (not shipped): produces valid streams that exercise every branch of the library's
decoder -- CONSTANT / VERBATIM / FIXED / LPC subframes, 4- and 5-bit Rice parameters, partitioned residuals,
escape partitions, wasted bits, independent / left-side / side-right / mid-side stereo, fixed and variable block
sizes, 8..32 bits per sample -- with correct CRC-8 / CRC-16 and the STREAMINFO MD5.  The format is the public
FLAC specification; nothing here comes from the reference (it has no codec code at all)."""
import hashlib

import numpy as np


def _crc_table(poly, bits):
    top = 1 << (bits - 1)
    mask = (1 << bits) - 1
    tab = []
    for i in range(256):
        c = i << (bits - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
        tab.append(c)
    return tab


_CRC8, _CRC16 = _crc_table(0x07, 8), _crc_table(0x8005, 16)


def crc8(data):
    c = 0
    for b in data:
        c = _CRC8[c ^ b]
    return c


def crc16(data):
    c = 0
    for b in data:
        c = ((c << 8) & 0xFFFF) ^ _CRC16[(c >> 8) ^ b]
    return c


class BitWriter:
    def __init__(self):
        self.chunks = []          # list of uint8 bit arrays (values 0/1)

    def put(self, value, nbits):
        if nbits:
            v = int(value) & ((1 << nbits) - 1)
            self.chunks.append(np.array([(v >> (nbits - 1 - i)) & 1 for i in range(nbits)], dtype=np.uint8))

    def put_bits(self, bit_array):
        self.chunks.append(np.asarray(bit_array, dtype=np.uint8))

    def unary(self, n):
        self.put_bits(np.concatenate((np.zeros(n, dtype=np.uint8), [1])))

    def nbits(self):
        return sum(len(c) for c in self.chunks)

    def align(self):
        pad = (-self.nbits()) % 8
        if pad:
            self.put_bits(np.zeros(pad, dtype=np.uint8))

    def tobytes(self):
        bits = np.concatenate(self.chunks) if self.chunks else np.zeros(0, dtype=np.uint8)
        assert len(bits) % 8 == 0
        return np.packbits(bits).tobytes()


def _rice_bits(res, k):
    """Vectorised Rice code (parameter k) of signed residuals -> 0/1 array."""
    res = np.asarray(res, dtype=np.int64)
    u = np.where(res >= 0, 2 * res, -2 * res - 1).astype(np.int64)       # zig-zag
    q = u >> k
    lens = q + 1 + k
    ends = np.cumsum(lens)
    starts = ends - lens
    bits = np.zeros(int(ends[-1]) if len(res) else 0, dtype=np.uint8)
    bits[starts + q] = 1                                                 # unary terminator
    for b in range(k):                                                   # k low bits, MSB first
        bits[starts + q + 1 + b] = (u >> (k - 1 - b)) & 1
    return bits


def _best_k(res, kmax):
    mean = float(np.mean(np.abs(res))) if len(res) else 0.0
    k = int(np.ceil(np.log2(mean + 1))) if mean > 0 else 0
    return min(max(k, 0), kmax)


def _residual(bw, res, blocksize, order, method=0, porder=0, escape_first=False):
    pbits = 4 if method == 0 else 5
    bw.put(method, 2)
    bw.put(porder, 4)
    parts = 1 << porder
    pos = 0
    for part in range(parts):
        cnt = (blocksize >> porder) - (order if part == 0 else 0)
        seg = res[pos:pos + cnt]
        pos += cnt
        if escape_first and part == 0:
            nb = int(max(1, max(int(np.abs(seg).max(initial=0)).bit_length() + 1, 1)))
            bw.put((1 << pbits) - 1, pbits)
            bw.put(nb, 5)
            for v in seg:
                bw.put(int(v), nb)
        else:
            k = _best_k(seg, (1 << pbits) - 2)
            bw.put(k, pbits)
            bw.put_bits(_rice_bits(seg, k))


_FIXED = {0: (), 1: (1,), 2: (2, -1), 3: (3, -3, 1), 4: (4, -6, 4, -1)}


def _predict_residual(x, coefs, shift=0):
    x = np.asarray(x, dtype=np.int64)
    order = len(coefs)
    pred = np.zeros(len(x) - order, dtype=np.int64)
    for i, c in enumerate(coefs):
        pred += c * x[order - 1 - i:len(x) - 1 - i]
    return x[order:] - (pred >> shift)


def _subframe(bw, x, bps, kind, order=2, wasted=0, method=0, porder=0, escape_first=False):
    x = np.asarray(x, dtype=np.int64)
    if wasted:
        assert not np.any(x & ((1 << wasted) - 1))
        x = x >> wasted
        bps -= wasted
    n = len(x)
    bw.put(0, 1)
    code = {"constant": 0, "verbatim": 1, "fixed": 8 + order, "lpc": 32 + order - 1}[kind]
    bw.put(code, 6)
    bw.put(1 if wasted else 0, 1)
    if wasted:
        bw.unary(wasted - 1)
    if kind == "constant":
        bw.put(int(x[0]), bps)
    elif kind == "verbatim":
        for v in x:
            bw.put(int(v), bps)
    elif kind == "fixed":
        for v in x[:order]:
            bw.put(int(v), bps)
        _residual(bw, _predict_residual(x, _FIXED[order]), n, order, method, porder, escape_first)
    else:                                               # LPC: the fixed predictor's taps scaled by 2^shift
        shift, prec = 3, 8
        coefs = [c << shift for c in _FIXED[min(order, 4)]] + [0] * max(0, order - 4)
        for v in x[:order]:
            bw.put(int(v), bps)
        bw.put(prec - 1, 4)
        bw.put(shift, 5)
        for c in coefs:
            bw.put(c, prec)
        _residual(bw, _predict_residual(x, coefs, shift), n, order, method, porder, escape_first)


def _utf8_number(v):
    """FLAC's extended UTF-8 coding of a frame / sample number (up to 36 bits in 7 bytes)."""
    if v < 0x80:
        return bytes([v])
    for nb in range(2, 8):
        if v < 1 << ((7 - nb) + 6 * (nb - 1)):
            lead = ((0xFF << (8 - nb)) & 0xFF) | (v >> (6 * (nb - 1)))
            return bytes([lead] + [0x80 | ((v >> (6 * i)) & 0x3F) for i in range(nb - 2, -1, -1)])
    raise ValueError("number too large")


def encode_flac(pcm, sr, bps, blocksize=4096, stereo="independent", kind="fixed", order=2, variable=False, wasted=0,
                method=0, porder=0, escape_first=False, vary_blocks=False):
    """pcm: int array (frames, channels) -> bytes of a FLAC stream."""
    pcm = np.asarray(pcm, dtype=np.int64)
    if pcm.ndim == 1:
        pcm = pcm[:, None]
    frames, ch = pcm.shape
    width = (bps + 7) // 8
    raw = pcm.astype("<i8").reshape(-1, 1).view(np.uint8)[:, :width].tobytes()
    body = bytearray()
    pos, number, sizes = 0, 0, []
    while pos < frames:
        bs = blocksize
        if vary_blocks and number % 2:
            bs = max(16, blocksize // 2)
        bs = min(bs, frames - pos)
        blk = pcm[pos:pos + bs]
        hdr = bytearray([0xFF, 0xF8 | (1 if variable else 0)])
        hdr.append((7 << 4) | 0)                               # 16-bit block size follows; sample rate from STREAMINFO
        assign = {"independent": ch - 1, "left_side": 8, "right_side": 9, "mid_side": 10}[stereo]
        ss = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}.get(bps, 0)
        hdr.append((assign << 4) | (ss << 1))
        hdr += _utf8_number(pos if variable else number)
        hdr += bytes([(bs - 1) >> 8, (bs - 1) & 0xFF])
        hdr.append(crc8(hdr))
        bw = BitWriter()
        args = dict(kind=kind, order=order, wasted=wasted, method=method, porder=porder, escape_first=escape_first)
        if len(blk) <= order or (porder and (bs >> porder) << porder != bs) or (porder and (bs >> porder) <= order):
            args.update(kind="verbatim", porder=0)
        if stereo == "independent":
            for c in range(ch):
                _subframe(bw, blk[:, c], bps, **args)
        else:
            L, R = blk[:, 0], blk[:, 1]
            if stereo == "left_side":
                _subframe(bw, L, bps, **args)
                _subframe(bw, L - R, bps + 1, **{**args, "wasted": 0})
            elif stereo == "right_side":
                _subframe(bw, L - R, bps + 1, **{**args, "wasted": 0})
                _subframe(bw, R, bps, **args)
            else:
                _subframe(bw, (L + R) >> 1, bps, **{**args, "wasted": 0})
                _subframe(bw, L - R, bps + 1, **{**args, "wasted": 0})
        bw.align()
        frame = bytes(hdr) + bw.tobytes()
        frame += crc16(frame).to_bytes(2, "big")
        body += frame
        sizes.append(bs)
        pos += bs
        number += 1
    min_b = max_b = blocksize
    if vary_blocks:
        min_b, max_b = min(sizes[:-1] or sizes), max(sizes)
    v = (sr << 44) | ((ch - 1) << 41) | ((bps - 1) << 36) | frames
    info = min_b.to_bytes(2, "big") + max_b.to_bytes(2, "big") + bytes(6) + v.to_bytes(8, "big") + hashlib.md5(raw).digest()
    padding = bytes([0x81, 0, 0, 8]) + bytes(8)                          # a last metadata block after STREAMINFO
    return b"fLaC" + bytes([0x00, 0, 0, 34]) + info + padding + bytes(body)
