// Shared by all kernels of libcombo_avs_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "combo_avs.h"  // the public C ABI (include/combo_avs.h)

#include <atomic>

#define COMBO_WAVE 64

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of a function ON ONE DEVICE: a launcher that opted in once per
// process (`static bool`) launches its > 64 KiB-LDS kernel without the opt-in on every other device the process touches, and the
// launch fails.  One bit per device ordinal, set after a successful opt-in on the current device; atomic, because launches may
// come from several host threads (setting the attribute twice is harmless).
struct ComboDevFlag {
  std::atomic<unsigned long long> bits{0ull};
  static unsigned long long bit() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return 1ull << (dev & 63);
  }
  bool is_set() const { return (bits.load(std::memory_order_acquire) & bit()) != 0ull; }
  void mark() { bits.fetch_or(bit(), std::memory_order_release); }
};

// Workgroups are dispatched round-robin over the 8 XCDs (blockIdx % 8), each with a private L2.  Maps blockIdx to a
// logical index such that every XCD owns a contiguous range of the n logical indices (bijective): neighbours in the
// logical order share an L2 and run at the same time.
__device__ __forceinline__ int xcd_contiguous(int id, int n) {
  const int q = n >> 3, r = n & 7, xcd = id & 7, j = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}


// Index arithmetic of the element-wise kernels: a 64-bit integer division costs ~100+ vector instructions on CDNA (no hardware
// divider), more than the whole body of a bandwidth-bound kernel - the depth-wise convolution ran at 1.2 TB/s because of three
// of them per thread.  Every tensor here has < 2^32 elements, so: 32-bit division whenever the dividend fits (the 64-bit path
// stays for safety).
__device__ __forceinline__ long long fast_div(long long a, int b) {
  return ((unsigned long long)a >> 32) == 0 ? (long long)((unsigned)a / (unsigned)b) : a / b;
}
__device__ __forceinline__ int fast_mod(long long a, int b) {
  return ((unsigned long long)a >> 32) == 0 ? (int)((unsigned)a % (unsigned)b) : (int)(a % b);
}

int combo_cu_limit(void);  // abi.hip: 0 = no limit; else the CU budget of the persistent GEMM launches of this host thread

// ---- device-side launch timing (works inside a replayed hipGraph, where HIP refuses event records) ----
// A slot = COMBO_TS_SLOT_U64 x u64: 16 sub-slots, one 128-byte line each, {earliest workgroup start, latest workgroup end}
// in wall-clock ticks; sub-slot 0 also carries {sum of durations, launches} at +2 / +3.  The host hands every instrumented
// launch the next slot of the buffer given to combo_timing_set_buffer (a graph node keeps its slot over all replays) and
// remembers (kind, work) per slot.  A workgroup costs two FIRE-AND-FORGET atomics (min of the start, max of the end) on the
// sub-slot blockIdx % 16; combo_timing_fold (one tiny launch per step, timing.hip) reduces the sub-slots, adds end - start
// to the sum and re-arms the slot.  Measured against rocprofv3: atomics of one launch on ONE address serialise at ~15-20 ns
// each, returning or not - a "last workgroup adds the duration" protocol cost 25 us on a 1280-workgroup launch, and plain
// min / max on a single pair of words still 18 us; spread over 16 lines it is ~1 us.
enum { COMBO_TS_MSDA_FWD = 0, COMBO_TS_GEMM_F32 = 1, COMBO_TS_GEMM_X3 = 2, COMBO_TS_GEMM_TN = 3, COMBO_TS_ATTN_FWD = 4,
       COMBO_TS_ATTN_BWD = 5, COMBO_TS_MSDA_BWD = 6, COMBO_TS_BIFUSE = 7, COMBO_TS_GEMM_BF16 = 8, COMBO_TS_CONV_WGRAD = 9, COMBO_TS_KINDS = 10 };
enum { COMBO_TS_SUBS = 16, COMBO_TS_SUB_U64 = 16, COMBO_TS_SLOT_U64 = 256 };
// host; nullptr when timing is off (timing.hip).  work: flops (bytes for the HBM-bound kinds); bytes: algorithmic HBM bytes
// of the launch (operands read once + result written once) for the kinds that report a second, HBM-side fraction
unsigned long long* combo_timing_next_slot(int kind, double work, double bytes = 0.0);

__device__ __forceinline__ unsigned long long* combo_ts_sub(unsigned long long* ts) {
  return ts + ((blockIdx.x + 5u * blockIdx.y) & (COMBO_TS_SUBS - 1)) * COMBO_TS_SUB_U64;
}
__device__ __forceinline__ void combo_ts_begin(unsigned long long* ts) {
  if (ts && threadIdx.x == 0) atomicMin(combo_ts_sub(ts), (unsigned long long)wall_clock64());
}
__device__ __forceinline__ void combo_ts_end(unsigned long long* ts) {
  if (!ts) return;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(combo_ts_sub(ts) + 1, (unsigned long long)wall_clock64());
}
