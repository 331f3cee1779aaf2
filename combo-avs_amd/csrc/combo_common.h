// Shared by all kernels of libcombo_avs_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "combo_avs.h"  // the public C ABI (include/combo_avs.h)

#define COMBO_WAVE 64
