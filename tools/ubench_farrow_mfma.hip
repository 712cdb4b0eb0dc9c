// settle the MFMA question with a number.
//
// The fc = 1 half of K_sinc (unity path, csrc/sinc.hip taps_unity_ct) spends 156 of its 209 VALU instructions per output
// on the taps n = 5 .. 31, whose weights are polynomials in q = shift^2 with FIXED coefficients:
//     far(s) = s (e0 + q e1 + q^2 e2) + (d0 + q d1 + q^2 d2),
//     e_k = sum_n c_k[n] (x[c+n] + x[c-n]),   d_k = sum_n n c_k[n] (x[c+n] - x[c-n])        (c_0 = A, c_1 = B, c_2 = C)
// i.e. six fixed 63-tap FIR filters evaluated on the INPUT grid (a Farrow bank).  This micro-benchmark builds that
// bank on the matrix cores inside the wave structure of K_sinc (a wave = 256 outputs = 4 rows of 64, its own LDS span):
//     D[(filter f, position i)][block b] += A[(f, i)][k] B[k][b]        v_mfma_f32_16x16x32_f16
//     B[k][b] = x16[8 b + k]            the signal: TWO rows (128 outputs) re-based so that block b = 8 window centres
//                                       starts on a multiple of 8 halves -> 16-byte aligned ds_read_b128 (an 8-byte or
//                                       2-byte aligned one costs 256 cycles instead of 30: tools/exp/lds_pattern.hip)
//     A[(f, i)][k] = coefficient of tap n = k - 31 - i of filter f: constants, 2 filters x 8 positions per M tile, ten
//                                       fragments in a 10 KB LDS table shared by the workgroup
//     D: a lane holds four consecutive positions of ONE filter -> one ds_write_b128 per tile
// float16 has 11 significant bits, so signal and the two dominant filters are split hi + lo * 2^-12 (3 products where
// they matter): (e0 d0)h, (e1 d1)h, (e0 d0)l, (e2 d2)h x xh and (e0 d0)h, (e1 d1)h x xl; only the K slices that hold
// non-zero taps are run  ->  15 MFMAs per 128 outputs, 30 per wave.  The bank (6 floats per position) lands in LDS, every
// output lane gathers the six values of its window centre and finishes with 5 FMAs.
// (Earlier attempts, in git history: signal as the A operand read with 2-byte-aligned ds_read_b128 -- numerically
// identical, 3.6x SLOWER than the VALU loops; blocks of 4 positions with 8-byte-aligned reads -- 1.6x slower.)
//
// Compared with the VALU form of the same sums (the product's literal-FMA loops, copied), both inside the same staging
// code, on a +-0 % "speed 0.99" read head (every wave on the unity path), for three signals; accuracy against a float64
// evaluation with the EXACT R_n(q) = (win_n/pi)/(n^2 - q).  `full = 1` adds the n = 1..4 reciprocal taps and the centre
// tap (VALU in both forms) so the ratio is that of the whole unity tap path.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pyaudiorestoration_amd/csrc tools/ubench_farrow_mfma.hip -o tools/ubench_farrow_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>
#include "sinc_taps_gen.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NT = 32, kWave = 64, kOut = 4, kWaveOut = 256, kMargin = 37, kTileCap = 384;
constexpr int kRowBuf = 224;                            // halves of a two-row image: 128 centres + 63 taps + K padding
constexpr int kFrags = 10;                              // constant A fragments (1 KB each)
using T = par::TapTab<NT>;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const float lds_cfloat;

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ float sinpi_half(float x) {
  const float z = x * x;
  float p = -0.00737043094f;
  p = fmaf(p, z, 0.0821458866f);
  p = fmaf(p, z, -0.599264529f);
  p = fmaf(p, z, 2.55016404f);
  p = fmaf(p, z, -5.16771278f);
  p = fmaf(p, z, 3.14159265f);
  return p * x;
}

// placement of the ubench: p_j = 0.99 j + 40.25  (period 0.99 <= 1: fc = 1)
__device__ __host__ __forceinline__ double pos_of(long long j) { return 0.99 * (double)j + 40.25; }

// VALU form of the far taps for two outputs of a lane (the product's loop body, n >= n_from)
template <int N_FROM, int R>
__device__ __forceinline__ void far_valu(const float* __restrict__ tile, const int (&c)[R], const float (&q)[R], const int nt_rt,
                                         float (&e)[R], float (&d)[R]) {
  float e0[R], e1[R], e2[R], d0[R], d1[R], d2[R];
  lds_cfloat* base[R];
  lds_cfloat* tl = (lds_cfloat*)tile;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    e0[r] = e1[r] = e2[r] = d0[r] = d1[r] = d2[r] = 0.0f;
    base[r] = tl + c[r] - NT;
  }
  static_for<(NT + 3) / 4>([&](auto cidx) {
    constexpr int n0 = decltype(cidx)::value * 4 + 1;
    if (nt_rt >= n0) static_for<(n0 + 3 <= NT ? 4 : NT - n0 + 1)>([&](auto idx) {
      constexpr int n = n0 + decltype(idx)::value;
      constexpr int mode = T::mode[n];
      constexpr float fn = (float)n;
      if constexpr (n >= N_FROM && n < NT) {
        if constexpr (n == T::p1_from) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            e1[r] = fmaf(q[r], e2[r], e1[r]);
            d1[r] = fmaf(q[r], d2[r], d1[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float sp = base[r][NT + n], sm = base[r][NT - n];
          const float D = sp - sm, E = sp + sm;
          if constexpr (mode == 0) {
#pragma clang fp contract(off)
            const float x = q[r] * T::B[n] + T::A[n];
            const float Rn = __builtin_amdgcn_rcpf(x);
            const float DR = D * Rn;
            if constexpr (n & 1) {
              e0[r] = fmaf(-E, Rn, e0[r]);
              d0[r] = fmaf(DR, -fn, d0[r]);
            } else {
              e0[r] = fmaf(E, Rn, e0[r]);
              d0[r] = fmaf(DR, fn, d0[r]);
            }
          } else if constexpr (mode == 1) {
            e0[r] = fmaf(E, T::A[n], e0[r]);
            e1[r] = fmaf(E, T::B[n], e1[r]);
            e2[r] = fmaf(E, T::C[n], e2[r]);
            d0[r] = fmaf(D, fn * T::A[n], d0[r]);
            d1[r] = fmaf(D, fn * T::B[n], d1[r]);
            d2[r] = fmaf(D, fn * T::C[n], d2[r]);
          } else if constexpr (mode == 2) {
            e0[r] = fmaf(E, T::A[n], e0[r]);
            e1[r] = fmaf(E, T::B[n], e1[r]);
            d0[r] = fmaf(D, fn * T::A[n], d0[r]);
            d1[r] = fmaf(D, fn * T::B[n], d1[r]);
          } else {
            e0[r] = fmaf(E, T::A[n], e0[r]);
            d0[r] = fmaf(D, fn * T::A[n], d0[r]);
          }
        }
      }
    });
  });
#pragma unroll
  for (int r = 0; r < R; ++r) {
    e[r] = fmaf(q[r], e1[r], e0[r]);
    d[r] = fmaf(q[r], d1[r], d0[r]);
  }
}

// the reciprocal taps n = 1 .. 4 and the centre tap (VALU in both forms): returns centre + s e + d of those taps
template <int R>
__device__ __forceinline__ void near_valu(const float* __restrict__ tile, const int (&c)[R], const float (&s)[R], const float (&q)[R],
                                          float (&acc)[R]) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float e0 = 0.0f, d0 = 0.0f;
    static_for<4>([&](auto idx) {
      constexpr int n = 1 + decltype(idx)::value;
      constexpr float fn = (float)n;
      const float sp = tile[c[r] + n], sm = tile[c[r] - n];
      const float D = sp - sm, E = sp + sm;
      const float x = fmaf(q[r], T::B[n], T::A[n]);
      const float Rn = __builtin_amdgcn_rcpf(x);
      const float DR = D * Rn;
      if constexpr (n & 1) {
        e0 = fmaf(-E, Rn, e0);
        d0 = fmaf(DR, -fn, d0);
      } else {
        e0 = fmaf(E, Rn, e0);
        d0 = fmaf(DR, fn, d0);
      }
    });
    const float centre = tile[c[r]] * __builtin_amdgcn_rcpf(s[r] * T::B[0]);
    acc[r] = centre + fmaf(s[r], e0, d0);
  }
}

struct BFrags {           // the four constant B fragments, per lane 8 halves each
  half8 b1[2], b2[2];
};

// MODE 0: VALU far taps; MODE 1: MFMA far taps.  FULL: add the near taps + the -sinpi(s) factor (the whole unity path).
template <int MODE, int FULL>
__global__ __launch_bounds__(256, MODE ? 5 : 6) void k_far(const float* __restrict__ sig, long long len_in, long long n_out,
                                                const half8* __restrict__ btab, float* __restrict__ out, int nt_rt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int kWaveBytes = kTileCap * 4 + (MODE ? 6 * 128 * 4 : 0);      // the bank overwrites the float16 image
  const int l = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  unsigned char* wbase = lds_raw + (MODE ? kFrags * 1024 : 0) + wv * kWaveBytes;
  if (MODE == 1) {                                   // the constant A fragments: [fragment][lane] 16 bytes
    // the product's own table (csrc/sinc_taps_gen.h, tools/gen_sinc_taps.py); UB_HOST_TABLE=1 passes the host-built copy instead
    const half8* tab = btab ? btab : reinterpret_cast<const half8*>(par::kFarrowFrags32);
    for (int i = threadIdx.x; i < kFrags * 64; i += blockDim.x) reinterpret_cast<half8*>(lds_raw)[i] = tab[i];
    __syncthreads();
  }
  float* tile = reinterpret_cast<float*>(wbase);
  const long long jw = ((long long)blockIdx.x * 4 + wv) * kWaveOut;
  if (jw + kWaveOut > n_out) return;
  // placement (cheap stand-in for the records)
  int c[kOut];
  float s[kOut], q[kOut];
  const long long anchor = (long long)rint(pos_of(jw)) & ~1ll;
#pragma unroll
  for (int r = 0; r < kOut; ++r) {
    const double p = pos_of(jw + l + 64 * r) - (double)anchor;
    const double rf = rint(p);
    c[r] = (int)rf;
    const float sh = (float)(p - rf);
    s[r] = sh == 0.0f ? 1e-20f : sh;
    q[r] = s[r] * s[r];
  }
  const int mn = __builtin_amdgcn_readlane(c[0], 0), mx = __builtin_amdgcn_readlane(c[kOut - 1], 63);
  const int nspan = mx - mn + 2 * kMargin + 1;
  const long long lo = anchor + mn - kMargin;
  for (int i = l; i < nspan; i += kWave) tile[i] = sig[lo + i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int r = 0; r < kOut; ++r) c[r] = c[r] - mn + kMargin;          // LDS index of the window centre
  float res[kOut];
  if (MODE == 0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ca[2] = {c[2 * h], c[2 * h + 1]};
      const float qa[2] = {q[2 * h], q[2 * h + 1]};
      float ea[2], da[2];
      far_valu<5, 2>(tile, ca, qa, nt_rt, ea, da);
      res[2 * h] = fmaf(s[2 * h], ea[0], da[0]);
      res[2 * h + 1] = fmaf(s[2 * h + 1], ea[1], da[1]);
    }
  } else {
    // per row PAIR: float16 image (hi, lo x 4096) of the 224 samples its windows touch, the bank, the gather
    _Float16* rb = reinterpret_cast<_Float16*>(wbase + kTileCap * 4);                 // [2][224] halves
    float* bank = reinterpret_cast<float*>(wbase + kTileCap * 4);                     // [6][128] floats (after the image is consumed)
    const unsigned rb_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(wbase + kTileCap * 4);
    const unsigned ca_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_raw + (unsigned)l * 16u;
    const int bb = l & 15, g = l >> 4;
#pragma unroll
    for (int rp = 0; rp < kOut / 2; ++rp) {
      const int p0 = __builtin_amdgcn_readlane(c[2 * rp], 0);          // first centre of the pair (LDS index); centres p0 .. p0 + 127
      for (int i = l; i < kRowBuf; i += kWave) {
        const int ti = p0 - 31 + i;
        const float x = ti < nspan ? tile[ti] : 0.0f;
        const _Float16 h = (_Float16)x;
        rb[i] = h;
        rb[kRowBuf + i] = (_Float16)((x - (float)h) * 4096.0f);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // fragment element k = 32 ks + 8 g + j  ->  x16[8 bb + k]
      const unsigned off = rb_addr + (unsigned)(8 * bb + 8 * g) * 2u;
      half8 xh[3], xl[3];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(xh[ks]) : "v"(off + (unsigned)(64 * ks)));
        asm volatile("ds_read_b128 %0, %1" : "=v"(xl[ks]) : "v"(off + (unsigned)(64 * ks + 2 * kRowBuf)));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xl[0]), "+v"(xl[1]), "+v"(xl[2]) :: "memory");
      float4v a_e0 = {0.0f, 0.0f, 0.0f, 0.0f}, a_e1 = a_e0, a_l0 = a_e0, a_e2 = a_e0, a_x0 = a_e0, a_x1 = a_e0;
      // constants: fragments 0-2 (e0 d0)h slices 0-2; 3-4 (e1 d1)h slices 0-1; 5-7 (e0 d0)l slices 0-2; 8-9 (e2 d2)h slices 0-1
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        half8 c0, cl, c1, c2;
        asm volatile("ds_read_b128 %0, %1" : "=v"(c0) : "v"(ca_addr + (unsigned)(ks * 1024)));
        asm volatile("ds_read_b128 %0, %1" : "=v"(cl) : "v"(ca_addr + (unsigned)((5 + ks) * 1024)));
        if (ks < 2) {
          asm volatile("ds_read_b128 %0, %1" : "=v"(c1) : "v"(ca_addr + (unsigned)((3 + ks) * 1024)));
          asm volatile("ds_read_b128 %0, %1" : "=v"(c2) : "v"(ca_addr + (unsigned)((8 + ks) * 1024)));
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(cl), "+v"(c1), "+v"(c2) :: "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(cl) :: "memory");
        }
        a_e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c0, xh[ks], a_e0, 0, 0, 0);
        a_l0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl, xh[ks], a_l0, 0, 0, 0);
        a_x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c0, xl[ks], a_x0, 0, 0, 0);
        if (ks < 2) {
          a_e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1, xh[ks], a_e1, 0, 0, 0);
          a_e2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c2, xh[ks], a_e2, 0, 0, 0);
          a_x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1, xl[ks], a_x1, 0, 0, 0);
        }
      }
      // D[row = 4 g + reg][col = bb]: row m = (filter m >> 3, position m & 7): g = 0,1 -> first filter of the tile, positions
      // 4 (g & 1) + reg of block bb; g = 2,3 -> second filter.  Bank layout [6 filters][128 positions].
      constexpr float lo_w = 1.0f / 4096.0f;
      const float4v v0 = a_e0 + (a_l0 + a_x0) * lo_w, v1 = a_e1 + a_x1 * lo_w;
      const int fsel = g >> 1, pos4 = 8 * bb + 4 * (g & 1);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         // every lane has read the image the bank overwrites
      __builtin_amdgcn_wave_barrier();
      *reinterpret_cast<float4v*>(bank + (0 + fsel) * 128 + pos4) = v0;      // e0 | d0
      *reinterpret_cast<float4v*>(bank + (2 + fsel) * 128 + pos4) = v1;      // e1 | d1
      *reinterpret_cast<float4v*>(bank + (4 + fsel) * 128 + pos4) = a_e2;    // e2 | d2
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * rp + rr;
        const float* f = bank + (c[r] - p0);
        const float e = fmaf(q[r], fmaf(q[r], f[4 * 128], f[2 * 128]), f[0]), d = fmaf(q[r], fmaf(q[r], f[5 * 128], f[3 * 128]), f[128]);
        res[r] = fmaf(s[r], e, d) * (1.0f / 32.0f);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         // the image is rewritten by the next pair
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  if (FULL) {
    float nr[kOut];
    near_valu<kOut>(tile, c, s, q, nr);
#pragma unroll
    for (int r = 0; r < kOut; ++r) res[r] = -sinpi_half(s[r]) * (nr[r] + res[r]);
  }
#pragma unroll
  for (int r = 0; r < kOut; ++r) out[jw + l + 64 * r] = res[r];
}

// ---- host ---------------------------------------------------------------------------------------------------------------
static double win_n(int n) { return (double)(float)(0.5 + 0.5 * cos(M_PI * (double)n / (double)NT)); }

int main(int argc, char** argv) {
  const long long n_out = (argc > 1 ? atoll(argv[1]) : 64ll << 20) / 1024 * 1024;
  const long long len_in = n_out + 4096;
  // A fragments.  M tile rows m = (filter m >> 3, position i = m & 7); lane: row m = lane & 15, g = lane >> 4; element j of
  // slice ks: tap n = 32 ks + 8 g + j - 31 - i.  Fragments 0-2 (e0 d0)h, 3-4 (e1 d1)h, 5-7 (e0 d0) lo x 4096, 8-9 (e2 d2)h.
  std::vector<_Float16> btab(kFrags * 64 * 8);
  auto coef = [&](int f, int n) -> double {                 // exact float32 literal the VALU form uses, as a double, x 32
    const int a = abs(n);
    if (a < 5 || a >= NT) return 0.0;
    const double sg = n < 0 ? -1.0 : 1.0;
    double v = 0.0;
    switch (f) {
      case 0: v = T::A[a]; break;                                              // e0
      case 1: v = T::mode[a] <= 2 ? T::B[a] : 0.0; break;                      // e1
      case 2: v = T::mode[a] == 1 ? T::C[a] : 0.0; break;                      // e2
      case 3: v = sg * (double)(float)((float)a * T::A[a]); break;             // d0
      case 4: v = T::mode[a] <= 2 ? sg * (double)(float)((float)a * T::B[a]) : 0.0; break;
      case 5: v = T::mode[a] == 1 ? sg * (double)(float)((float)a * T::C[a]) : 0.0; break;
    }
    return 32.0 * v;
  };
  struct Frag { int fe, fd, ks, lo; };
  const Frag frags[kFrags] = {{0, 3, 0, 0}, {0, 3, 1, 0}, {0, 3, 2, 0}, {1, 4, 0, 0}, {1, 4, 1, 0},
                              {0, 3, 0, 1}, {0, 3, 1, 1}, {0, 3, 2, 1}, {2, 5, 0, 0}, {2, 5, 1, 0}};
  for (int fr = 0; fr < kFrags; ++fr)
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 8; ++j) {
        const int m = lane & 15, i = m & 7, g = lane >> 4, n = 32 * frags[fr].ks + 8 * g + j - 31 - i;
        const double cf = coef((m >> 3) ? frags[fr].fd : frags[fr].fe, n);
        btab[(fr * 64 + lane) * 8 + j] = frags[fr].lo ? (_Float16)((cf - (double)(_Float16)cf) * 4096.0) : (_Float16)cf;
      }
  // every non-zero tap must sit in a slice that is run: (e1 d1) and (e2 d2) skip slice 2
  for (int f : {1, 2, 4, 5})
    for (int i = 0; i < 8; ++i)
      for (int k = 64; k < 96; ++k)
        if (coef(f, k - 31 - i) != 0.0) { printf("coefficient of filter %d in a skipped slice\n", f); return 1; }
  half8* d_btab;
  float *d_sig, *d_out;
  CHECK(hipMalloc(&d_btab, btab.size() * 2));
  CHECK(hipMemcpy(d_btab, btab.data(), btab.size() * 2, hipMemcpyHostToDevice));
  if (!getenv("UB_HOST_TABLE")) {                      // default: the generated table compiled into the product
    CHECK(hipFree(d_btab));
    d_btab = nullptr;
  }
  CHECK(hipMalloc(&d_sig, len_in * 4));
  CHECK(hipMalloc(&d_out, n_out * 4));
  std::vector<float> sig(len_in), outv(n_out);
  const char* names[3] = {"white noise (uniform +-1)", "full-scale Nyquist tone", "full-scale tone at 0.45 fs"};
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int wave_bytes1 = kTileCap * 4 + 6 * 128 * 4;
  printf("# tools/ubench_farrow_mfma: %lld outputs per launch, NT = 32 (taps n = 5..31 = 54 of 64), read head at speed 0.99 (fc = 1)\n", n_out);
  for (int full = 0; full < 2; ++full) {
    double ms[2] = {0, 0};
    for (int mode = 0; mode < 2; ++mode) {
      for (int sgi = 0; sgi < 3; ++sgi) {
        uint64_t st = 0x9E3779B97F4A7C15ull;
        for (long long i = 0; i < len_in; ++i) {
          if (sgi == 0) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            sig[i] = (float)((double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0);
          } else if (sgi == 1) sig[i] = (i & 1) ? -1.0f : 1.0f;
          else sig[i] = (float)cos(0.9 * M_PI * (double)i + 0.2);
        }
        CHECK(hipMemcpy(d_sig, sig.data(), len_in * 4, hipMemcpyHostToDevice));
        const dim3 grid((unsigned)(n_out / 1024)), block(256);
        const size_t lds = 4 * (mode ? wave_bytes1 : kTileCap * 4) + (mode ? kFrags * 1024 : 0);
        auto launch = [&]() {
          if (mode == 0 && !full) hipLaunchKernelGGL((k_far<0, 0>), grid, block, lds, 0, d_sig, len_in, n_out, d_btab, d_out, NT);
          if (mode == 0 && full) hipLaunchKernelGGL((k_far<0, 1>), grid, block, lds, 0, d_sig, len_in, n_out, d_btab, d_out, NT);
          if (mode == 1 && !full) hipLaunchKernelGGL((k_far<1, 0>), grid, block, lds, 0, d_sig, len_in, n_out, d_btab, d_out, NT);
          if (mode == 1 && full) hipLaunchKernelGGL((k_far<1, 1>), grid, block, lds, 0, d_sig, len_in, n_out, d_btab, d_out, NT);
        };
        launch();
        CHECK(hipDeviceSynchronize());
        if (sgi == 0) {
          CHECK(hipEventRecord(e0));
          for (int rep = 0; rep < 10; ++rep) launch();
          CHECK(hipEventRecord(e1));
          CHECK(hipEventSynchronize(e1));
          float t;
          CHECK(hipEventElapsedTime(&t, e0, e1));
          ms[mode] = t / 10;
        }
        CHECK(hipMemcpy(outv.data(), d_out, n_out * 4, hipMemcpyDeviceToHost));
        // float64 reference with the exact R_n(q) on a sample of outputs
        double worst = 0.0, peak = 0.0;
        int nbad = 0;
        if (getenv("UB_MAP") && mode == 1 && !full && sgi == 0) {
          for (long long j = 0; j < 512; ++j) {
            const double p = pos_of(j), cr = rint(p), sh = p - cr, qq = sh * sh;
            const long long ci = (long long)cr;
            double far = 0.0;
            for (int n = 5; n < NT; ++n) {
              const double Rn = (win_n(n) / M_PI) / ((double)n * n - qq) * ((n & 1) ? -1.0 : 1.0);
              far += Rn * (sh * ((double)sig[ci + n] + (double)sig[ci - n]) + n * ((double)sig[ci + n] - (double)sig[ci - n]));
            }
            printf("%c", fabs(outv[j] - far) < 1e-4 ? '.' : 'X');
            if (j % 64 == 63) printf("\n");
          }
        }
        for (long long j = 1000; j < n_out; j += 4099) {
          const double p = pos_of(j), cr = rint(p), sh = p - cr, qq = sh * sh;
          const long long ci = (long long)cr;
          double far = 0.0, nearv = 0.0;
          for (int n = 1; n < NT; ++n) {
            const double Rn = (win_n(n) / M_PI) / ((double)n * n - qq) * ((n & 1) ? -1.0 : 1.0);
            const double E = (double)sig[ci + n] + (double)sig[ci - n], D = (double)sig[ci + n] - (double)sig[ci - n];
            (n >= 5 ? far : nearv) += Rn * (sh * E + n * D);
          }
          const double centre = sh == 0.0 ? 0.0 : (double)sig[ci] * (-win_n(0) / (M_PI * sh));
          const double want = full ? (sh == 0.0 ? (double)sig[ci] * win_n(0) : -sin(M_PI * sh) * (centre + nearv + far)) : far;
          const double sc = full ? 1.0 : fabs(sin(M_PI * sh));        // what the far error becomes in the output
          if (getenv("UB_DEBUG") && j < 1000 + 4099 * 4) printf("  j=%lld ci=%lld sh=%.4f got=%.6g want=%.6g (far %.6g near %.6g centre %.6g)\n", j, ci, sh, (double)outv[j], want, far, nearv, centre);
          if (getenv("UB_DEBUG2") && fabs((double)outv[j] - want) * sc > 1e-3 && nbad++ < 12) printf("  BAD j=%lld (wave off %lld: r=%lld l=%lld) ci=%lld sh=%.4f got=%.6g want=%.6g\n", j, j % 256, (j % 256) / 64, j % 64, ci, sh, (double)outv[j], want);
          worst = fmax(worst, fabs((double)outv[j] - want) * sc);
          peak = fmax(peak, fabs(full ? want : 1.0));
        }
        printf("%s  %-5s  %-28s  max |err| / peak = %.2e\n", full ? "whole unity path (n = 0..31)" : "far taps only (n = 5..31)  ",
               mode ? "MFMA" : "VALU", names[sgi], worst / peak);
      }
    }
    printf("%s  VALU %.3f ms = %.2f ps/output   MFMA %.3f ms = %.2f ps/output   ratio %.2fx\n",
           full ? "whole unity path (n = 0..31)" : "far taps only (n = 5..31)  ", ms[0], ms[0] * 1e9 / n_out, ms[1], ms[1] * 1e9 / n_out,
           ms[0] / ms[1]);
  }
  return 0;
}
