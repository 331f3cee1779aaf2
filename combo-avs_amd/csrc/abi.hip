// Library introspection entry points (host only) and the event helpers bench.py times kernels with.
#include "combo_common.h"

// CU budget of the persistent GEMM kernels (gemm_nt3.hip, gemm_f32.hip: one workgroup per CU): 0 = the whole device.  A caller that
// runs two independent launch chains on two HIP streams (the Siam pair of backbones, meta_arch.MaskFormer.parallel_backbones) gives
// each chain half of the CUs - the chains then execute side by side instead of one launch after the other, and a launch's fixed
// costs (dispatch, ring priming, drain of the last stores: ~8 us) overlap with the other chain's streaming.  Host-side state, read
// at launch time; thread-local, because autograd issues the backward launches from its own thread.
static thread_local int g_cu_limit = 0;
int combo_cu_limit(void) { return g_cu_limit; }

extern "C" {
int combo_set_cu_limit(int n) {
  const int prev = g_cu_limit;
  g_cu_limit = n > 0 ? n : 0;
  return prev;
}
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
