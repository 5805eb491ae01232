// Library-level entry points of the C ABI (include/ococc_hip.h).
#include "common.hpp"

thread_local char ococc_err_buf[512] = {0};

extern "C" const char* ococc_last_error(void) { return ococc_err_buf; }
extern "C" int ococc_version(void) { return 100; /* round 1 */ }
extern "C" const char* ococc_arch(void) { return "gfx950"; }

extern "C" int ococc_timer_create(void** timer) {
  OCOCC_REQUIRE(timer, "null timer");
  hipEvent_t ev;
  OCOCC_HIP(hipEventCreate(&ev));
  *timer = (void*)ev;
  return OCOCC_OK;
}

extern "C" int ococc_timer_record(void* timer, int32_t in_graph, ococc_stream_t stream) {
  OCOCC_REQUIRE(timer, "null timer");
  OCOCC_HIP(hipEventRecordWithFlags((hipEvent_t)timer, (hipStream_t)stream,
                                    in_graph ? hipEventRecordExternal : hipEventRecordDefault));
  return OCOCC_OK;
}

extern "C" int ococc_timer_elapsed_ms(void* start, void* stop, float* ms) {
  OCOCC_REQUIRE(start && stop && ms, "null argument");
  OCOCC_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return OCOCC_OK;
}

extern "C" int ococc_timer_destroy(void* timer) {
  if (timer) OCOCC_HIP(hipEventDestroy((hipEvent_t)timer));
  return OCOCC_OK;
}
