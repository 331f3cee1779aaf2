// Host side of the device-side launch timing (combo_common.h): slot hand-out and the per-slot (kind, work) record.
#include <vector>

#include "combo_common.h"

namespace {
unsigned long long* g_base = nullptr;
int g_slots = 0, g_next = 0;
struct SlotInfo { int kind; double work; };
std::vector<SlotInfo> g_info;

__global__ void timing_fold_kernel(unsigned long long* base, int slots) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= slots) return;
  unsigned long long* ts = base + (long long)COMBO_TS_SLOT_U64 * i;
  unsigned long long t0 = ~0ull, t1 = 0ull;
  for (int s = 0; s < COMBO_TS_SUBS; ++s) {
    unsigned long long* sub = ts + s * COMBO_TS_SUB_U64;
    t0 = sub[0] < t0 ? sub[0] : t0;
    t1 = sub[1] > t1 ? sub[1] : t1;
    sub[0] = ~0ull;
    sub[1] = 0ull;
  }
  if (t1 != 0ull && t0 != ~0ull) {
    ts[2] += t1 - t0;
    ts[3] += 1ull;
  }
}
}  // namespace

unsigned long long* combo_timing_next_slot(int kind, double work) {
  if (!g_base || g_next >= g_slots) return nullptr;  // out of slots: later launches run untimed
  g_info[g_next] = SlotInfo{kind, work};
  return g_base + (long long)COMBO_TS_SLOT_U64 * g_next++;
}

extern "C" {

int combo_timing_set_buffer(void* buf, int slots) {
  g_base = reinterpret_cast<unsigned long long*>(buf);
  g_slots = buf ? slots : 0;
  g_next = 0;
  g_info.assign(g_slots > 0 ? g_slots : 0, SlotInfo{-1, 0.0});
  return 0;
}

int combo_timing_slots_used(void) { return g_next; }

int combo_timing_fold(combo_stream_t stream) {
  if (!g_base || g_next <= 0) return 0;
  hipLaunchKernelGGL(timing_fold_kernel, dim3((g_next + 255) / 256), dim3(256), 0, (hipStream_t)stream, g_base, g_next);
  return (int)hipGetLastError();
}

int combo_timing_slot_info(int slot, int* kind, double* work) {
  if (slot < 0 || slot >= g_next) return COMBO_EINVAL;
  if (kind) *kind = g_info[slot].kind;
  if (work) *work = g_info[slot].work;
  return 0;
}

int combo_wall_clock_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
  return khz;
}

}  // extern "C"
