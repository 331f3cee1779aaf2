// Library introspection entry points (host only) and the event helpers bench.py times kernels with.
#include "combo_common.h"

extern "C" {
int combo_abi_version(void) { return 3; }
const char* combo_build_arch(void) { return "gfx950"; }

// Timing events that also work inside a captured hipGraph: with external != 0 the record becomes an event-record NODE
// (hipEventRecordExternal), so after every replay the event holds that replay's timestamp.
int combo_event_create(void** event) {
  if (!event) return COMBO_EINVAL;
  hipEvent_t e;
  hipError_t rc = hipEventCreate(&e);
  *event = (void*)e;
  return (int)rc;
}
int combo_event_record(void* event, combo_stream_t stream, int external) {
  if (!event) return COMBO_EINVAL;
  return (int)hipEventRecordWithFlags((hipEvent_t)event, (hipStream_t)stream,
                                      external ? hipEventRecordExternal : hipEventRecordDefault);
}
int combo_event_elapsed_us(void* start, void* stop, float* us) {
  if (!start || !stop || !us) return COMBO_EINVAL;
  float ms = 0.f;
  hipError_t rc = hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop);
  *us = ms * 1000.f;
  return (int)rc;
}
int combo_event_destroy(void* event) { return event ? (int)hipEventDestroy((hipEvent_t)event) : COMBO_EINVAL; }
}
