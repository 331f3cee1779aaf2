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
