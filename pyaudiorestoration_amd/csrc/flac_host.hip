// Host-side FLAC decoder (no device code): the data-format side of the path.  The reference reads
// its inputs through soundfile/libsndfile (util/io_ops.py:7-16), which this image lacks; hour-long 192 kHz
// files are ~10^9 samples, so the decoder is native and frame-parallel rather than Python.
//
// Format subset = everything the FLAC format defines for PCM: CONSTANT / VERBATIM / FIXED(0-4) / LPC(1-32)
// subframes, Rice residuals (4- and 5-bit parameters, escape partitions), wasted bits, independent and
// left-side / side-right / mid-side stereo, fixed and variable block sizes, 4..32 bits per sample.
// Every frame is checked (header CRC-8, frame CRC-16); optionally the decoded PCM against STREAMINFO's MD5.
//
// Parallelism: FLAC frames are independently decodable.  The byte stream is cut into T ranges; each worker
// resynchronises on the first frame in its range whose header CRC-8 AND whole-frame CRC-16 hold, learns its
// absolute sample position from the frame header, and decodes up to the next worker's resync point.
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <exception>
#include <thread>
#include <vector>
#include "par_common.h"

namespace par {
namespace flac {

struct Info {
  int sr = 0, ch = 0, bps = 0, min_block = 0, max_block = 0;
  int64_t total = 0;
  uint8_t md5[16] = {0};
  size_t audio_off = 0;
};

static uint8_t g_crc8[256];
static uint16_t g_crc16[256];
static std::once_flag g_crc_once;
static void init_crc() {
  for (int i = 0; i < 256; ++i) {
    uint8_t c = (uint8_t)i;
    for (int k = 0; k < 8; ++k) c = (uint8_t)((c << 1) ^ ((c & 0x80) ? 0x07 : 0));
    g_crc8[i] = c;
    uint16_t d = (uint16_t)(i << 8);
    for (int k = 0; k < 8; ++k) d = (uint16_t)((d << 1) ^ ((d & 0x8000) ? 0x8005 : 0));
    g_crc16[i] = d;
  }
}

struct Bits {
  const uint8_t* p;
  size_t n, pos = 0;        // byte position of the next unread byte
  uint64_t acc = 0;         // bit reservoir, MSB-aligned in the low `have` bits
  int have = 0;
  bool bad = false;
  Bits(const uint8_t* data, size_t len, size_t start) : p(data), n(len), pos(start) {}
  inline void fill(int need) {
    while (have < need) {
      uint64_t b = 0;
      if (pos < n) b = p[pos]; else bad = true;
      ++pos;
      acc = (acc << 8) | b;
      have += 8;
    }
  }
  inline uint32_t u(int bits) {          // 0..32 bits
    if (bits == 0) return 0;
    fill(bits);
    have -= bits;
    return (uint32_t)((acc >> have) & ((bits == 32) ? 0xffffffffull : ((1ull << bits) - 1)));
  }
  inline int32_t s(int bits) {           // sign-extended, 1..32 bits
    if (bits == 0) return 0;
    const uint32_t v = u(bits);
    if (bits == 32) return (int32_t)v;
    return (int32_t)(v << (32 - bits)) >> (32 - bits);
  }
  inline uint32_t unary() {              // zeros before the next 1 (the 1 is consumed)
    uint32_t count = 0;
    for (;;) {
      if (have == 0) fill(8);
      if (bad) return count;
      const uint64_t window = acc & ((have == 64) ? ~0ull : ((1ull << have) - 1));
      if (window) {
        const int lead = __builtin_clzll(window) - (64 - have);
        have -= lead + 1;
        return count + (uint32_t)lead;
      }
      count += (uint32_t)have;
      have = 0;
    }
  }
  inline void align() { have -= have & 7; }
  inline size_t byte_pos() const { return pos - (size_t)(have >> 3); }    // valid when aligned
};

struct FrameHeader {
  int blocksize = 0, ch_assign = 0, bps = 0;
  bool variable = false;
  uint64_t number = 0;      // frame number (fixed) or first sample number (variable)
  size_t len = 0;           // header bytes incl. CRC-8
};

// Parses and CRC-checks a frame header at `off`; false when it is not a plausible header of THIS stream.
static bool parse_header(const uint8_t* d, size_t n, size_t off, const Info& in, FrameHeader* h) {
  if (off + 6 > n || d[off] != 0xFF || (d[off + 1] & 0xFE) != 0xF8) return false;
  h->variable = d[off + 1] & 1;
  const int bs_code = d[off + 2] >> 4, sr_code = d[off + 2] & 15;
  const int ch_code = d[off + 3] >> 4, ss_code = (d[off + 3] >> 1) & 7;
  if (d[off + 3] & 1) return false;
  if (bs_code == 0 || sr_code == 15 || ch_code > 10 || ss_code == 3) return false;      // 7 = 32 bits (RFC 9639)
  size_t q = off + 4;
  // UTF-8 style coded number (up to 36 bits)
  const uint8_t lead = d[q++];
  int follow = 0;
  uint64_t num = 0;
  if (lead < 0x80) {
    num = lead;
  } else {
    while (follow < 7 && (lead & (0x80 >> follow))) ++follow;
    if (follow < 2 || follow > 7) return false;
    num = lead & (0x7F >> follow);
    for (int i = 1; i < follow; ++i) {
      if (q >= n || (d[q] & 0xC0) != 0x80) return false;
      num = (num << 6) | (d[q++] & 0x3F);
    }
  }
  h->number = num;
  int bs = 0;
  if (bs_code == 1) bs = 192;
  else if (bs_code <= 5) bs = 576 << (bs_code - 2);
  else if (bs_code == 6) { if (q >= n) return false; bs = d[q++] + 1; }
  else if (bs_code == 7) { if (q + 1 >= n) return false; bs = ((d[q] << 8) | d[q + 1]) + 1; q += 2; }
  else bs = 256 << (bs_code - 8);
  if (sr_code == 12) q += 1;
  else if (sr_code == 13 || sr_code == 14) q += 2;
  if (q >= n) return false;
  uint8_t c = 0;
  for (size_t i = off; i < q; ++i) c = g_crc8[c ^ d[i]];
  if (c != d[q]) return false;
  static const int kBps[8] = {0, 8, 12, 0, 16, 20, 24, 32};
  h->bps = ss_code == 0 ? in.bps : kBps[ss_code];
  h->blocksize = bs;
  h->ch_assign = ch_code;
  h->len = q + 1 - off;
  const int nch = ch_code < 8 ? ch_code + 1 : 2;
  if (nch != in.ch || h->bps != in.bps) return false;
  if (in.max_block && bs > in.max_block) return false;
  return true;
}

static bool residual(Bits& br, int blocksize, int order, int32_t* out /* blocksize - order values */) {
  const uint32_t method = br.u(2);
  if (method > 1) return false;
  const int pbits = method == 0 ? 4 : 5;
  const int porder = (int)br.u(4);
  const int parts = 1 << porder;
  if ((blocksize >> porder) << porder != blocksize && porder != 0) return false;
  int idx = 0;
  for (int part = 0; part < parts; ++part) {
    int cnt = (blocksize >> porder) - (part == 0 ? order : 0);
    if (cnt < 0) return false;
    const uint32_t k = br.u(pbits);
    if (k == (uint32_t)((1 << pbits) - 1)) {
      const int nb = (int)br.u(5);
      for (int i = 0; i < cnt; ++i) out[idx++] = br.s(nb);
    } else {
      for (int i = 0; i < cnt; ++i) {
        const uint32_t q = br.unary();
        const uint32_t v = (q << k) | br.u((int)k);
        out[idx++] = (int32_t)(v >> 1) ^ -(int32_t)(v & 1);
      }
    }
    if (br.bad) return false;
  }
  return idx == blocksize - order;
}

static bool subframe(Bits& br, int blocksize, int bps, int64_t* s, int32_t* res) {
  if (br.u(1)) return false;
  const int typ = (int)br.u(6);
  int wasted = 0;
  if (br.u(1)) {
    wasted = (int)br.unary() + 1;
    bps -= wasted;
    if (bps < 1) return false;
  }
  auto rd = [&](int bits) -> int64_t {          // up to 33 bits (side channel of 32-bit audio)
    if (bits <= 32) return br.s(bits);
    const int64_t hi = br.s(bits - 32);
    return hi * 4294967296ll + (int64_t)br.u(32);
  };
  if (typ == 0) {
    const int64_t v = rd(bps);
    for (int i = 0; i < blocksize; ++i) s[i] = v;
  } else if (typ == 1) {
    for (int i = 0; i < blocksize; ++i) s[i] = rd(bps);
  } else if (typ >= 8 && typ <= 12) {
    const int order = typ - 8;
    if (order > blocksize) return false;
    for (int i = 0; i < order; ++i) s[i] = rd(bps);
    if (!residual(br, blocksize, order, res)) return false;
    const int32_t* r = res;
    switch (order) {
      case 0: for (int i = 0; i < blocksize; ++i) s[i] = *r++; break;
      case 1: for (int i = 1; i < blocksize; ++i) s[i] = *r++ + s[i - 1]; break;
      case 2: for (int i = 2; i < blocksize; ++i) s[i] = *r++ + 2 * s[i - 1] - s[i - 2]; break;
      case 3: for (int i = 3; i < blocksize; ++i) s[i] = *r++ + 3 * s[i - 1] - 3 * s[i - 2] + s[i - 3]; break;
      default: for (int i = 4; i < blocksize; ++i) s[i] = *r++ + 4 * s[i - 1] - 6 * s[i - 2] + 4 * s[i - 3] - s[i - 4];
    }
  } else if (typ >= 32) {
    const int order = (typ & 31) + 1;
    if (order > blocksize) return false;
    for (int i = 0; i < order; ++i) s[i] = rd(bps);
    const int prec = (int)br.u(4) + 1;
    if (prec == 16) return false;
    const int shift = br.s(5);
    if (shift < 0) return false;
    int64_t co[32];
    for (int i = 0; i < order; ++i) co[i] = br.s(prec);
    if (!residual(br, blocksize, order, res)) return false;
    for (int i = order; i < blocksize; ++i) {
      int64_t acc = 0;
      for (int j = 0; j < order; ++j) acc += co[j] * s[i - 1 - j];
      s[i] = res[i - order] + (acc >> shift);
    }
  } else {
    return false;
  }
  if (wasted)
    for (int i = 0; i < blocksize; ++i) s[i] = s[i] * ((int64_t)1 << wasted);
  return !br.bad;
}

// Decodes one frame starting at `off` into ch[c][0..blocksize).  Returns the byte length of the frame, 0 on failure.
struct Scratch {
  std::vector<int64_t> a, b;
  std::vector<int32_t> res;
  std::vector<std::vector<int64_t>> chans;
};
static size_t decode_frame(const uint8_t* d, size_t n, size_t off, const Info& in, FrameHeader* h, Scratch& sc) {
  if (!parse_header(d, n, off, in, h)) return 0;
  const int bs = h->blocksize;
  if ((int)sc.res.size() < bs) {
    sc.res.resize(bs);
    sc.a.resize(bs);
    sc.b.resize(bs);
  }
  if ((int)sc.chans.size() < in.ch) sc.chans.resize(in.ch);
  for (int c = 0; c < in.ch; ++c)
    if ((int)sc.chans[c].size() < bs) sc.chans[c].resize(bs);
  Bits br(d, n, off + h->len);
  if (h->ch_assign < 8) {
    for (int c = 0; c < in.ch; ++c)
      if (!subframe(br, bs, h->bps, sc.chans[c].data(), sc.res.data())) return 0;
  } else {
    const int bps0 = h->bps + (h->ch_assign == 9 ? 1 : 0), bps1 = h->bps + (h->ch_assign == 9 ? 0 : 1);
    if (!subframe(br, bs, bps0, sc.a.data(), sc.res.data())) return 0;
    if (!subframe(br, bs, bps1, sc.b.data(), sc.res.data())) return 0;
    int64_t* L = sc.chans[0].data();
    int64_t* R = sc.chans[1].data();
    if (h->ch_assign == 8) {                     // left, side = left - right
      for (int i = 0; i < bs; ++i) { L[i] = sc.a[i]; R[i] = sc.a[i] - sc.b[i]; }
    } else if (h->ch_assign == 9) {              // side, right
      for (int i = 0; i < bs; ++i) { L[i] = sc.a[i] + sc.b[i]; R[i] = sc.b[i]; }
    } else {                                     // mid, side
      for (int i = 0; i < bs; ++i) {
        const int64_t m = (sc.a[i] * 2) | (sc.b[i] & 1), sd = sc.b[i];
        L[i] = (m + sd) >> 1;
        R[i] = (m - sd) >> 1;
      }
    }
  }
  br.align();
  if (br.bad) return 0;
  const size_t end = br.byte_pos();
  if (end + 2 > n) return 0;
  uint16_t c = 0;
  for (size_t i = off; i < end; ++i) c = (uint16_t)((c << 8) ^ g_crc16[(c >> 8) ^ d[i]]);
  if (c != (uint16_t)((d[end] << 8) | d[end + 1])) return 0;
  return end + 2 - off;
}

static int read_info(const uint8_t* d, size_t n, Info* in) {
  if (n < 8 || memcmp(d, "fLaC", 4) != 0) return PAR_ERR_ARG;
  size_t pos = 4;
  bool got = false;
  for (;;) {
    if (pos + 4 > n) return PAR_ERR_ARG;
    const uint8_t hdr = d[pos];
    const size_t len = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
    if (pos + 4 + len > n) return PAR_ERR_ARG;
    if ((hdr & 0x7F) == 0 && len >= 34) {
      const uint8_t* b = d + pos + 4;
      in->min_block = (b[0] << 8) | b[1];
      in->max_block = (b[2] << 8) | b[3];
      uint64_t v = 0;
      for (int i = 10; i < 18; ++i) v = (v << 8) | b[i];
      in->sr = (int)(v >> 44);
      in->ch = (int)((v >> 41) & 7) + 1;
      in->bps = (int)((v >> 36) & 31) + 1;
      in->total = (int64_t)(v & ((1ull << 36) - 1));
      memcpy(in->md5, b + 18, 16);
      got = true;
    }
    pos += 4 + len;
    if (hdr & 0x80) break;
  }
  in->audio_off = pos;
  return got ? PAR_OK : PAR_ERR_ARG;
}

// ---- MD5 (RFC 1321) of the decoded PCM, for the STREAMINFO self-check --------------------------------------
struct Md5 {
  uint32_t h[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
  uint64_t len = 0;
  uint8_t buf[64];
  size_t fill = 0;
  static inline uint32_t rol(uint32_t x, int c) { return (x << c) | (x >> (32 - c)); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af,
        0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa,
        0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8,
        0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
        0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97,
        0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
        0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                              14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                              4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t w[16];
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    uint32_t a = h[0], b = h[1], c = h[2], dd = h[3];
    for (int i = 0; i < 64; ++i) {
      uint32_t f;
      int g;
      if (i < 16) { f = (b & c) | (~b & dd); g = i; }
      else if (i < 32) { f = (dd & b) | (~dd & c); g = (5 * i + 1) & 15; }
      else if (i < 48) { f = b ^ c ^ dd; g = (3 * i + 5) & 15; }
      else { f = c ^ (b | ~dd); g = (7 * i) & 15; }
      const uint32_t t = dd;
      dd = c;
      c = b;
      b = b + rol(a + f + K[i] + w[g], S[i]);
      a = t;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += dd;
  }
  void update(const uint8_t* p, size_t n) {
    len += n;
    if (fill) {
      const size_t take = std::min(n, 64 - fill);
      memcpy(buf + fill, p, take);
      fill += take; p += take; n -= take;
      if (fill == 64) { block(buf); fill = 0; }
    }
    for (; n >= 64; p += 64, n -= 64) block(p);
    if (n) { memcpy(buf, p, n); fill = n; }
  }
  void final(uint8_t out[16]) {
    const uint64_t bits = len * 8;
    const uint8_t pad = 0x80;
    update(&pad, 1);
    const uint8_t z = 0;
    while (fill != 56) update(&z, 1);
    uint8_t lb[8];
    for (int i = 0; i < 8; ++i) lb[i] = (uint8_t)(bits >> (8 * i));
    update(lb, 8);
    for (int i = 0; i < 4; ++i)
      for (int k = 0; k < 4; ++k) out[4 * i + k] = (uint8_t)(h[i] >> (8 * k));
  }
};

}  // namespace flac
}  // namespace par

extern "C" int par_flac_info(const void* data, size_t nbytes, int* sample_rate, int* channels, int* bits,
                             int64_t* total_frames, uint8_t* md5) {
  using namespace par;
  PAR_REQUIRE(data, PAR_ERR_ARG, "par_flac_info: null pointer");
  flac::Info in;
  const int rc = flac::read_info(static_cast<const uint8_t*>(data), nbytes, &in);
  PAR_REQUIRE(rc == PAR_OK, PAR_ERR_ARG, "par_flac_info: not a FLAC stream with a STREAMINFO block");
  if (sample_rate) *sample_rate = in.sr;
  if (channels) *channels = in.ch;
  if (bits) *bits = in.bps;
  if (total_frames) *total_frames = in.total;
  if (md5) memcpy(md5, in.md5, 16);
  return PAR_OK;
}

static int flac_decode_impl(const void* data, size_t nbytes, float* out, int64_t frames_cap, int n_threads, int verify_md5,
                            int64_t* frames_decoded) {
  using namespace par;
  PAR_REQUIRE(data && out && frames_decoded, PAR_ERR_ARG, "par_flac_decode_f32: null pointer");
  std::call_once(flac::g_crc_once, flac::init_crc);
  const uint8_t* d = static_cast<const uint8_t*>(data);
  flac::Info in;
  PAR_REQUIRE(flac::read_info(d, nbytes, &in) == PAR_OK, PAR_ERR_ARG, "par_flac_decode_f32: not a FLAC stream");
  PAR_REQUIRE(in.total > 0, PAR_ERR_UNSUPPORTED, "par_flac_decode_f32: STREAMINFO has no total sample count");
  PAR_REQUIRE(in.total <= frames_cap, PAR_ERR_WORKSPACE, "par_flac_decode_f32: %lld frames do not fit the output (%lld)",
              (long long)in.total, (long long)frames_cap);
  const bool fixed = in.min_block == in.max_block && in.min_block > 0;
  int T = n_threads <= 0 ? (int)std::thread::hardware_concurrency() : n_threads;
  T = std::max(1, std::min(T, 256));
  const size_t audio_len = nbytes - in.audio_off;
  if (audio_len < (size_t)T * (1u << 16)) T = std::max<size_t>(1, audio_len >> 16);
  const double scale = (double)((int64_t)1 << (in.bps - 1));
  const int width = (in.bps + 7) / 8;
  std::vector<uint8_t> pcm;                        // packed little-endian PCM for the MD5 check
  if (verify_md5) pcm.resize((size_t)in.total * in.ch * width);

  // resync point of every worker
  std::vector<size_t> start(T + 1, nbytes);
  std::vector<int64_t> first(T + 1, in.total);
  std::atomic<int> fail{0};
  auto sample_of = [&](const flac::FrameHeader& h) -> int64_t {
    return h.variable ? (int64_t)h.number : (int64_t)h.number * (fixed ? in.min_block : h.blocksize);
  };
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        flac::Scratch sc;
        flac::FrameHeader h;
        size_t off = in.audio_off + (t == 0 ? 0 : audio_len / T * t);
        for (; off + 16 <= nbytes; ++off) {
          if (d[off] != 0xFF || (d[off + 1] & 0xFE) != 0xF8) continue;
          if (t != 0 && !fixed && !(d[off + 1] & 1)) continue;   // variable stream: mid-stream position needs sample numbers
          if (flac::decode_frame(d, nbytes, off, in, &h, sc)) {
            start[t] = off;
            first[t] = sample_of(h);
            return;
          }
          if (t == 0) break;                       // the first frame must sit right after the metadata
        }
        if (t == 0) fail = 1;
      });
    for (auto& x : th) x.join();
  }
  PAR_REQUIRE(!fail, PAR_ERR_ARG, "par_flac_decode_f32: no valid frame after the metadata blocks");
  // drop workers that found nothing / resynced past a later worker
  std::vector<int> live;
  for (int t = 0; t < T; ++t)
    if (start[t] < nbytes && (live.empty() || start[t] > start[live.back()])) live.push_back(t);
  std::atomic<int64_t> done{0};
  {
    std::vector<std::thread> th;
    for (size_t w = 0; w < live.size(); ++w)
      th.emplace_back([&, w] {
        flac::Scratch sc;
        flac::FrameHeader h;
        size_t off = start[live[w]];
        const size_t stop = w + 1 < live.size() ? start[live[w + 1]] : nbytes;
        int64_t at = first[live[w]];
        int64_t n_done = 0;
        while (off < stop && at < in.total) {
          const size_t len = flac::decode_frame(d, nbytes, off, in, &h, sc);
          // the header's own position is only meaningful for fixed-size or sample-numbered frames
          if (!len || ((fixed || h.variable) && sample_of(h) != at)) { fail = 2; return; }
          const int64_t take = std::min<int64_t>(h.blocksize, in.total - at);
          for (int c = 0; c < in.ch; ++c) {
            const int64_t* s = sc.chans[c].data();
            float* o = out + at * in.ch + c;
            for (int64_t i = 0; i < take; ++i) o[i * in.ch] = (float)((double)s[i] / scale);
            if (verify_md5) {
              uint8_t* p = pcm.data() + ((size_t)at * in.ch + c) * width;
              for (int64_t i = 0; i < take; ++i)
                for (int b = 0; b < width; ++b) p[(size_t)i * in.ch * width + b] = (uint8_t)((uint64_t)s[i] >> (8 * b));
            }
          }
          at += take;
          n_done += take;
          off += len;
        }
        if (off != stop && at < in.total && w + 1 < live.size()) fail = 3;   // did not land on the next worker's frame
        done += n_done;
      });
    for (auto& x : th) x.join();
  }
  PAR_REQUIRE(!fail, PAR_ERR_ARG, "par_flac_decode_f32: corrupt stream (frame CRC / sequence check failed, code %d)", (int)fail);
  PAR_REQUIRE(done == in.total, PAR_ERR_ARG, "par_flac_decode_f32: decoded %lld of %lld frames", (long long)done.load(),
              (long long)in.total);
  if (verify_md5) {
    bool any = false;
    for (int i = 0; i < 16; ++i) any = any || in.md5[i];
    if (any) {
      flac::Md5 m;
      m.update(pcm.data(), pcm.size());
      uint8_t dig[16];
      m.final(dig);
      PAR_REQUIRE(memcmp(dig, in.md5, 16) == 0, PAR_ERR_ARG, "par_flac_decode_f32: decoded PCM does not match the STREAMINFO MD5");
    }
  }
  *frames_decoded = in.total;
  return PAR_OK;
}

extern "C" int par_flac_decode_f32(const void* data, size_t nbytes, float* out, int64_t frames_cap, int n_threads, int verify_md5,
                                   int64_t* frames_decoded) {
  try {
    int rc = flac_decode_impl(data, nbytes, out, frames_cap, n_threads, verify_md5, frames_decoded);
    // a chance false resync (a byte pattern passing CRC-8 and CRC-16 inside another frame) only breaks the parallel
    // split, not the stream: decode serially before calling the file corrupt
    if (rc == PAR_ERR_ARG && n_threads != 1) rc = flac_decode_impl(data, nbytes, out, frames_cap, 1, verify_md5, frames_decoded);
    return rc;
  } catch (const std::exception& e) {       // e.g. bad_alloc for the MD5 staging of a stream that claims 2^36 frames
    par::set_error("par_flac_decode_f32: %s", e.what());
    return PAR_ERR_WORKSPACE;
  }
}
