// Host side of the device-side launch timing (combo_common.h): slot hand-out and the per-slot (kind, work) record.
#include <vector>

#include "combo_common.h"

namespace {
unsigned long long* g_base = nullptr;
int g_slots = 0, g_next = 0, g_high = 0;  // g_high: slots handed out so far (g_next rewinds in eager runs)
bool g_truncated = false;
struct SlotInfo { int kind; double work; double bytes; };
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

unsigned long long* combo_timing_next_slot(int kind, double work, double bytes) {
  if (!g_base) return nullptr;
  if (g_next >= g_slots) { g_truncated = true; return nullptr; }  // out of slots: later launches run untimed (reported)
  g_info[g_next] = SlotInfo{kind, work, bytes};
  unsigned long long* p = g_base + (long long)COMBO_TS_SLOT_U64 * g_next++;
  if (g_next > g_high) g_high = g_next;
  return p;
}

extern "C" {

int combo_timing_set_buffer(void* buf, int slots) {
  g_base = reinterpret_cast<unsigned long long*>(buf);
  g_slots = buf ? slots : 0;
  g_next = g_high = 0;
  g_truncated = false;
  g_info.assign(g_slots > 0 ? g_slots : 0, SlotInfo{-1, 0.0, 0.0});
  return 0;
}

int combo_timing_slots_used(void) { return g_high; }

// Eager (un-captured) steps: call before every step so that the step's i-th instrumented launch takes slot i again (a graph
// node keeps its slot by construction; without the rewind an eager run takes fresh slots every step and runs out).
int combo_timing_rewind(void) { g_next = 0; return 0; }

// 1 when a launch asked for a slot after the buffer ran out: the per-kind figures then cover only the slotted launches.
int combo_timing_truncated(void) { return g_truncated ? 1 : 0; }

int combo_timing_fold(combo_stream_t stream) {
  if (!g_base || g_high <= 0) return 0;
  hipLaunchKernelGGL(timing_fold_kernel, dim3((g_high + 255) / 256), dim3(256), 0, (hipStream_t)stream, g_base, g_high);
  return (int)hipGetLastError();
}

int combo_timing_slot_info(int slot, int* kind, double* work) {
  if (slot < 0 || slot >= g_high) return COMBO_EINVAL;
  if (kind) *kind = g_info[slot].kind;
  if (work) *work = g_info[slot].work;
  return 0;
}

int combo_timing_slot_bytes(int slot, double* bytes) {
  if (slot < 0 || slot >= g_high || !bytes) return COMBO_EINVAL;
  *bytes = g_info[slot].bytes;
  return 0;
}

int combo_wall_clock_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
  return khz;
}

}  // extern "C"
