// Host side of the device-side launch timing (combo_common.h): slot hand-out and the per-slot (kind, work) record.
#include <vector>

#include "combo_common.h"

namespace {
unsigned long long* g_base = nullptr;
int g_slots = 0, g_next = 0;
struct SlotInfo { int kind; double work; };
std::vector<SlotInfo> g_info;
}  // namespace

unsigned long long* combo_timing_next_slot(int kind, double work) {
  if (!g_base || g_next >= g_slots) return nullptr;  // out of slots: later launches run untimed
  g_info[g_next] = SlotInfo{kind, work};
  return g_base + 4LL * g_next++;
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
