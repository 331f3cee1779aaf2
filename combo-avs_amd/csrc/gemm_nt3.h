// Internal (not part of the C ABI): the launcher of csrc/gemm_nt3.hip, called by the combo_gemm_nt_x3_* / combo_conv3x3_nhwc_x3_*
// entry points.
#pragma once
#include "combo_avs.h"

struct combo_nt3_conv {
  int H, W, Cin;  // input map and channels
  int tap_step;   // split convolution: batch entry b computes taps b * tap_step .. (b + 1) * tap_step - 1 (K = tap_step * Cin); else 0
  int stride;     // 0 / 1: output map = input map; 2: output (ceil(H / 2), ceil(W / 2)) - the GEMM's M counts OUTPUT tokens
  int pad;        // 1: 3x3 taps around the (strided) centre, zero padding; 0: the single tap of a strided 1x1 convolution
};

// C_b[M,N] = A_b[M,K] . Bimg_b[N,K]^T (+ bias) (+ ReLU) (mask_b > 0 ? . : 0), `batch` problems at base + b * stride (elements; the
// mask shares C's pitch and stride; add: C += add before the ReLU, same layout); products: 3 = fp32-accurate split, 1 = plain bf16; conv != nullptr: implicit-GEMM 3x3
// convolution (A = NHWC tokens, K = 9 * Cin); force_cfg: 0 auto, 1 wide (256 x 128), 2 mid (128 x 128), 3 skinny (64 x 64), 4 tall (256 x 64).
int combo_nt3_launch(const float* A, long long lda, const float* Bimg, long long ldb, const float* bias, const float* mask, float* C,
                     long long ldc, long long M, int N, int K, int relu, int products, int batch, long long sA, long long sB,
                     long long sC, const combo_nt3_conv* conv, int force_cfg, combo_stream_t stream, const float* add = nullptr);
