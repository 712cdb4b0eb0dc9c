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

int par_version(void) { return 101; }

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
int par_stream_sync(int device, void* stream) {
  PAR_HIP_CHECK(hipSetDevice(device));
  PAR_HIP_CHECK(hipStreamSynchronize(par::as_stream(stream)));
  return PAR_OK;
}

}  // extern "C"
