// Shared by all kernels of libcombo_avs_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "combo_avs.h"  // the public C ABI (include/combo_avs.h)

#define COMBO_WAVE 64

// Workgroups are dispatched round-robin over the 8 XCDs (blockIdx % 8), each with a private L2.  Maps blockIdx to a
// logical index such that every XCD owns a contiguous range of the n logical indices (bijective): neighbours in the
// logical order share an L2 and run at the same time.
__device__ __forceinline__ int xcd_contiguous(int id, int n) {
  const int q = n >> 3, r = n & 7, xcd = id & 7, j = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// ---- device-side launch timing (works inside a replayed hipGraph, where HIP refuses event records) ----
// A slot = 4 x u64 {earliest workgroup start, workgroups done, sum of durations, launches} in wall-clock ticks.  The host
// hands every instrumented launch the next slot of the buffer given to combo_timing_set_buffer (a graph node keeps its slot
// over all replays) and remembers (kind, work) per slot; the last workgroup to finish adds (its end - the earliest start).
enum { COMBO_TS_MSDA_FWD = 0, COMBO_TS_GEMM_F32 = 1, COMBO_TS_GEMM_X3 = 2, COMBO_TS_GEMM_TN = 3, COMBO_TS_ATTN_FWD = 4,
       COMBO_TS_ATTN_BWD = 5, COMBO_TS_MSDA_BWD = 6, COMBO_TS_KINDS = 8 };
unsigned long long* combo_timing_next_slot(int kind, double work);  // host; nullptr when timing is off (timing.hip)

__device__ __forceinline__ void combo_ts_begin(unsigned long long* ts) {
  if (ts && threadIdx.x == 0) atomicMin(&ts[0], (unsigned long long)wall_clock64());
}
__device__ __forceinline__ void combo_ts_end(unsigned long long* ts) {
  if (!ts) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t1 = wall_clock64();
    __threadfence();
    if (atomicAdd(&ts[1], 1ull) == (unsigned long long)gridDim.x * gridDim.y * gridDim.z - 1ull) {
      const unsigned long long t0 = atomicExch(&ts[0], ~0ull);  // (also re-arms the slot for the next replay)
      atomicExch(&ts[1], 0ull);
      atomicAdd(&ts[2], t1 - t0);
      atomicAdd(&ts[3], 1ull);
    }
  }
}
