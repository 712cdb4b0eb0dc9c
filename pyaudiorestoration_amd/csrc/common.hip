// Library-level entry points: errors, device query, HIP events on caller streams.
#include "par_common.h"
#include <stdarg.h>

namespace par {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace par

extern "C" {

int par_version(void) { return 106; }     // 106 (r06): par_varispeed_fused_batch_f32 (several files per launch); 105 (r06): par_fused_redo_list; par_speed_to_pos_fill_fused NaN-fills from a plan without checkpoints (lazy / not fused_ok); the mono streaming launch is two kernels by stream kind (k_sinc_pipe<1, 2>, <1, 1>); 104 (r05): stereo form of the streaming kernel (interleaved NT = 32 files), end tiles inside the streaming launch; 103 (r05): lazy plans (fused_ok 2, force_host | 8), the streaming kernel is par_varispeed_fused_f32's default for mono NT = 32 (par_varispeed_fused_alone_f32 is gone), par_sosfiltfilt_batch_f64

int par_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int par_last_error(char* buf, int n) {
  if (!buf || n <= 0) return PAR_ERR_ARG;
  strncpy(buf, par::g_err, n - 1);
  buf[n - 1] = 0;
  return PAR_OK;
}

int par_event_create(void** ev) {
  PAR_REQUIRE(ev, PAR_ERR_ARG, "par_event_create: null");
  hipEvent_t e;
  PAR_HIP_CHECK(hipEventCreate(&e));
  *ev = e;
  return PAR_OK;
}
int par_event_destroy(void* ev) {
  PAR_HIP_CHECK(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)));
  return PAR_OK;
}
int par_event_record(void* ev, void* stream) {
  PAR_HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), par::as_stream(stream)));
  return PAR_OK;
}
int par_event_elapsed_ms(void* start, void* stop, float* ms) {
  PAR_REQUIRE(ms, PAR_ERR_ARG, "par_event_elapsed_ms: null");
  PAR_HIP_CHECK(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
  PAR_HIP_CHECK(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
  return PAR_OK;
}
int par_stream_create(int device, int low_priority, int cu_count, void** stream) {
  PAR_REQUIRE(stream, PAR_ERR_ARG, "par_stream_create: null");
  PAR_HIP_CHECK(hipSetDevice(device));
  hipStream_t st;
  if (cu_count != 0) {
    // cu_count > 0: the first cu_count compute units of the device's numbering; < 0: all BUT the first -cu_count
    // (hipExtStreamCreateWithCUMask: bit i of the mask = CU i)
    hipDeviceProp_t prop;
    PAR_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    const int total = prop.multiProcessorCount;
    const int n = cu_count > 0 ? cu_count : -cu_count;
    PAR_REQUIRE(n <= total && (cu_count > 0 || n < total), PAR_ERR_ARG, "par_stream_create: cu_count outside the device's compute units");
    std::vector<uint32_t> mask((total + 31) / 32, 0u);
    for (int i = 0; i < total; ++i)
      if ((i < n) == (cu_count > 0)) mask[i >> 5] |= 1u << (i & 31);
    PAR_HIP_CHECK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
  } else {
    int least = 0, greatest = 0;
    PAR_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    PAR_HIP_CHECK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, low_priority ? least : 0));
  }
  *stream = st;
  return PAR_OK;
}
int par_stream_destroy(void* stream) {
  PAR_HIP_CHECK(hipStreamDestroy(par::as_stream(stream)));
  return PAR_OK;
}
int par_stream_sync(int device, void* stream) {
  PAR_HIP_CHECK(hipSetDevice(device));
  PAR_HIP_CHECK(hipStreamSynchronize(par::as_stream(stream)));
  return PAR_OK;
}

}  // extern "C"
