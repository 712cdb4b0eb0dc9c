// Varispeed resample of one channel from a finished plan: positions (K_pos fill) then interpolation
// (K_sinc), optionally chunked so that the fill of chunk c+1 runs on a side stream while K_sinc works on
// chunk c (measured: no gain on MI355X, see below; default is one chunk).  Same results as par_speed_to_pos_fill + par_sinc_resample_f32 (same kernels, the
// chunk boundaries are K_sinc tile boundaries); this is the "higher-level slot" of resampling.run
// (reference util/resampling.py:184, :225-227).
#include "par_common.h"
#include "pos_plan.h"
#include <map>
#include <vector>

namespace par {

constexpr int kMaxChunks = 32;

struct Pipe {
  hipStream_t side = nullptr;
  hipEvent_t start = nullptr;
  hipEvent_t filled[kMaxChunks] = {};
  // optional profiling of the K_sinc launches (bench.py's roofline leg)
  bool profile = false;
  hipEvent_t t0[kMaxChunks] = {}, t1[kMaxChunks] = {};
  int timed_launches = 0;
  int64_t timed_samples = 0;
  std::vector<float> ms;          // per-launch durations collected by par_profile_read
};
static std::mutex g_pipe_mu;
static std::map<int, Pipe> g_pipes;

static int get_pipe(int device, Pipe** out) {
  std::lock_guard<std::mutex> lk(g_pipe_mu);
  Pipe& p = g_pipes[device];
  if (!p.side) {
    PAR_HIP_CHECK(hipStreamCreateWithFlags(&p.side, hipStreamNonBlocking));
    PAR_HIP_CHECK(hipEventCreateWithFlags(&p.start, hipEventDisableTiming));
    for (int i = 0; i < kMaxChunks; ++i) {
      PAR_HIP_CHECK(hipEventCreateWithFlags(&p.filled[i], hipEventDisableTiming));
      PAR_HIP_CHECK(hipEventCreate(&p.t0[i]));
      PAR_HIP_CHECK(hipEventCreate(&p.t1[i]));
    }
  }
  *out = &p;
  return PAR_OK;
}

}  // namespace par

extern "C" {

int par_varispeed_resample_f32(int device, const double* speeds, int64_t m, const void* work, int64_t len_out,
                               double* pos, const float* sig, int64_t sig_stride, int64_t len_in, int NT, float* out,
                               int64_t out_stride, int n_chunks, void* stream) {
  using namespace par;
  PAR_REQUIRE(speeds && work && pos && sig && out && m >= 2, PAR_ERR_ARG, "par_varispeed_resample_f32: null pointer");
  PAR_REQUIRE(len_out >= 2, PAR_ERR_ARG, "par_varispeed_resample_f32: len_out=%lld < 2", (long long)len_out);
  PAR_REQUIRE(NT >= 1 && NT <= 512 && len_in >= 1 && sig_stride >= 1 && out_stride >= 1, PAR_ERR_ARG,
              "par_varispeed_resample_f32: bad sizes");
  PAR_HIP_CHECK(hipSetDevice(device));
  hipStream_t main = as_stream(stream);
  // auto = 1: on MI355X the two kernels do not overlap usefully (measured, 115 M and 691 M outputs:
  // 1 chunk 1.91 / 10.42 ms per step, 4-32 chunks 1.96-2.15 / 10.67 ms) -- each launch already fills
  // all 256 CUs, so the side-stream fill only steals slots from K_sinc.  The chunked form is kept for
  // callers that want the first output samples early.
  if (n_chunks <= 0) n_chunks = 1;
  if (n_chunks > kMaxChunks) n_chunks = kMaxChunks;
  int64_t per = ceil_div(ceil_div(len_out, n_chunks), kSincTileOutputs) * kSincTileOutputs;
  n_chunks = (int)ceil_div(len_out, per);
  Pipe* p = nullptr;
  int rc = get_pipe(device, &p);
  if (rc != PAR_OK) return rc;
  std::lock_guard<std::mutex> lk(g_pipe_mu);      // one pipelined call per process at a time (shared side stream)
  // side stream starts after everything already queued on the caller's stream (the plan kernels)
  PAR_HIP_CHECK(hipEventRecord(p->start, main));
  PAR_HIP_CHECK(hipStreamWaitEvent(p->side, p->start, 0));
  for (int c = 0; c < n_chunks; ++c) {
    const int64_t j_lo = (int64_t)c * per;
    const int64_t j_hi = (c == n_chunks - 1) ? INT64_MAX : j_lo + per;
    rc = launch_pos_fill(speeds, m, work, pos, len_out, j_lo, j_hi, p->side);
    if (rc != PAR_OK) return rc;
    PAR_HIP_CHECK(hipEventRecord(p->filled[c], p->side));
  }
  for (int c = 0; c < n_chunks; ++c) {
    // outputs [j_lo, j_hi) read pos[j_hi] too: the first position of the next chunk's first segment
    const int need = c + 1 < n_chunks ? c + 1 : c;
    PAR_HIP_CHECK(hipStreamWaitEvent(main, p->filled[need], 0));
    const int64_t j_lo = (int64_t)c * per;
    const int64_t cnt = (j_lo + per <= len_out) ? per : len_out - j_lo;
    if (p->profile) PAR_HIP_CHECK(hipEventRecord(p->t0[c], main));
    rc = launch_sinc(device, pos, len_out, j_lo, cnt, sig, sig_stride, len_in, NT, out, out_stride, main);
    if (rc != PAR_OK) return rc;
    if (p->profile) PAR_HIP_CHECK(hipEventRecord(p->t1[c], main));
  }
  if (p->profile) {
    p->timed_launches = n_chunks;
    p->timed_samples = len_out;
  }
  return PAR_OK;
}

// Fused form: no position array at all.  Needs a plan made by par_speed_to_pos_plan_fused (checkpoints + tile map
// in `aux`); K_sinc regenerates each tile's float64 positions in LDS, bit-identical to the materialised path.
static int varispeed_fused_impl(const char* who, int device, const double* speeds, int64_t m, const void* work, const void* aux,
                                int64_t max_out, int64_t len_out, const float* sig, const float* sig1, int64_t sig_stride,
                                int64_t len_in, int NT, float* out, float* out1, int64_t out_stride, void* stream, int form = -1) {
  using namespace par;
  PAR_REQUIRE(speeds && work && aux && sig && out && m >= 2, PAR_ERR_ARG, "%s: null pointer", who);
  PAR_REQUIRE(len_out >= 2 && len_out <= max_out, PAR_ERR_ARG, "%s: len_out=%lld outside [2, max_out=%lld]", who,
              (long long)len_out, (long long)max_out);
  PAR_REQUIRE(NT >= 1 && NT <= 512 && len_in >= 1 && sig_stride >= 1 && out_stride >= 1, PAR_ERR_ARG, "%s: bad sizes", who);
  PAR_HIP_CHECK(hipSetDevice(device));
  Pipe* p = nullptr;
  int rc = get_pipe(device, &p);
  if (rc != PAR_OK) return rc;
  hipStream_t main = as_stream(stream);
  if (p->profile) PAR_HIP_CHECK(hipEventRecord(p->t0[0], main));
  rc = launch_sinc_fused(device, speeds, m, work, aux, max_out, len_out, sig, sig1, sig_stride, len_in, NT, out, out1,
                         out_stride, main, form);
  if (rc != PAR_OK) return rc;
  if (p->profile) {
    PAR_HIP_CHECK(hipEventRecord(p->t1[0], main));
    p->timed_launches = 1;
    p->timed_samples = len_out;
  }
  return PAR_OK;
}

int par_varispeed_fused_f32(int device, const double* speeds, int64_t m, const void* work, const void* aux,
                            int64_t max_out, int64_t len_out, const float* sig, int64_t sig_stride, int64_t len_in,
                            int NT, float* out, int64_t out_stride, void* stream) {
  return varispeed_fused_impl("par_varispeed_fused_f32", device, speeds, m, work, aux, max_out, len_out, sig, nullptr,
                              sig_stride, len_in, NT, out, nullptr, out_stride, stream);
}

int par_fused_redo_tiles(int device, const void* aux, int64_t max_out, int64_t m, int* tiles, void* stream) {
  using namespace par;
  PAR_REQUIRE(aux && tiles && m >= 2 && max_out >= 2, PAR_ERR_ARG, "par_fused_redo_tiles: bad argument");
  PAR_HIP_CHECK(hipSetDevice(device));
  const FusedAux av = fused_aux_view(const_cast<void*>(aux), max_out, m);
  PAR_HIP_CHECK(hipMemcpyAsync(tiles, av.redo_count, sizeof(int), hipMemcpyDeviceToHost, as_stream(stream)));
  PAR_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
  return PAR_OK;
}

// ... and which: the first min(count, cap) entries of that list (tile indices, in the order the streams pushed them)
int par_fused_redo_list(int device, const void* aux, int64_t max_out, int64_t m, int* tiles, int cap, int* count, void* stream) {
  using namespace par;
  PAR_REQUIRE(aux && count && (tiles || cap == 0) && cap >= 0 && m >= 2 && max_out >= 2, PAR_ERR_ARG, "par_fused_redo_list: bad argument");
  PAR_HIP_CHECK(hipSetDevice(device));
  const FusedAux av = fused_aux_view(const_cast<void*>(aux), max_out, m);
  PAR_HIP_CHECK(hipMemcpyAsync(count, av.redo_count, sizeof(int), hipMemcpyDeviceToHost, as_stream(stream)));
  PAR_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
  const int n = *count < cap ? *count : cap;
  if (n > 0) {
    PAR_HIP_CHECK(hipMemcpyAsync(tiles, av.redo_list, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, as_stream(stream)));
    PAR_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
  }
  return PAR_OK;
}

// Two channels of one file in one launch (same positions, same strides): position regeneration, prologue and tap
// weights are computed once for both.
int par_varispeed_fused_stereo_f32(int device, const double* speeds, int64_t m, const void* work, const void* aux,
                                   int64_t max_out, int64_t len_out, const float* sig0, const float* sig1,
                                   int64_t sig_stride, int64_t len_in, int NT, float* out0, float* out1,
                                   int64_t out_stride, void* stream) {
  PAR_REQUIRE(sig1 && out1, PAR_ERR_ARG, "par_varispeed_fused_stereo_f32: null pointer");
  return varispeed_fused_impl("par_varispeed_fused_stereo_f32", device, speeds, m, work, aux, max_out, len_out, sig0, sig1,
                              sig_stride, len_in, NT, out0, out1, out_stride, stream);
}

// Several planned files in one call (r06): the K_sinc launches of files that all take the streaming kernel in one form (NT = 32;
// mono on unit strides, or interleaved stereo) are merged -- their tails and launch gaps are paid once per batch of up to eight;
// any other mix is done file by file.  Results are bit-identical to par_varispeed_fused_f32 / _stereo_f32 per file.
int par_varispeed_fused_batch_f32(int device, int n_items, const par_fused_item* items, int NT, void* stream) {
  using namespace par;
  PAR_REQUIRE(n_items >= 0 && (items || n_items == 0), PAR_ERR_ARG, "par_varispeed_fused_batch_f32: null items");
  PAR_REQUIRE(NT >= 1 && NT <= 512, PAR_ERR_ARG, "par_varispeed_fused_batch_f32: NT=%d outside [1,512]", NT);
  if (n_items == 0) return PAR_OK;
  std::vector<FusedBatchItem> v((size_t)n_items);
  for (int k = 0; k < n_items; ++k) {
    const par_fused_item& f = items[k];
    PAR_REQUIRE(f.speeds && f.work && f.aux && f.sig0 && f.out0 && f.m >= 2, PAR_ERR_ARG, "par_varispeed_fused_batch_f32: item %d: null pointer", k);
    PAR_REQUIRE((f.sig1 == nullptr) == (f.out1 == nullptr), PAR_ERR_ARG, "par_varispeed_fused_batch_f32: item %d: sig1 / out1 come together", k);
    PAR_REQUIRE(f.len_out >= 2 && f.len_out <= f.max_out, PAR_ERR_ARG, "par_varispeed_fused_batch_f32: item %d: len_out=%lld outside [2, max_out=%lld]", k,
                (long long)f.len_out, (long long)f.max_out);
    PAR_REQUIRE(f.len_in >= 1 && f.sig_stride >= 1 && f.out_stride >= 1, PAR_ERR_ARG, "par_varispeed_fused_batch_f32: item %d: bad sizes", k);
    v[k] = FusedBatchItem{f.speeds, f.m, f.work, f.aux, f.max_out, f.len_out, f.sig0, f.sig1, f.sig_stride, f.len_in, f.out0, f.out1, f.out_stride};
  }
  PAR_HIP_CHECK(hipSetDevice(device));
  return launch_sinc_fused_batch(device, n_items, v.data(), NT, as_stream(stream));
}

// profiling hook for bench.py: HIP-event timing of the K_sinc launches issued by the last pipelined call
int par_profile_enable(int device, int on) {
  using namespace par;
  PAR_HIP_CHECK(hipSetDevice(device));
  Pipe* p = nullptr;
  int rc = get_pipe(device, &p);
  if (rc != PAR_OK) return rc;
  p->profile = on != 0;
  p->timed_launches = 0;
  return PAR_OK;
}

// sums the K_sinc launch durations of the last pipelined call (synchronises on them)
int par_profile_read(int device, float* total_ms, int* launches, int64_t* samples) {
  using namespace par;
  PAR_REQUIRE(total_ms && launches && samples, PAR_ERR_ARG, "par_profile_read: null");
  PAR_HIP_CHECK(hipSetDevice(device));
  Pipe* p = nullptr;
  int rc = get_pipe(device, &p);
  if (rc != PAR_OK) return rc;
  float tot = 0.0f;
  for (int c = 0; c < p->timed_launches; ++c) {
    float ms = 0.0f;
    PAR_HIP_CHECK(hipEventSynchronize(p->t1[c]));
    PAR_HIP_CHECK(hipEventElapsedTime(&ms, p->t0[c], p->t1[c]));
    tot += ms;
  }
  *total_ms = tot;
  *launches = p->timed_launches;
  *samples = p->timed_samples;
  return PAR_OK;
}

}  // extern "C"
