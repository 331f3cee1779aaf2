// Internal (not part of the C ABI): the launcher of csrc/gemm_nt3.hip, called by the combo_gemm_nt_x3_* / combo_conv3x3_nhwc_x3_*
// entry points.
#pragma once
#include "combo_avs.h"

// fp16 pieces: the B image (the weight) is split from 2^COMBO_F16_BSCALE_LOG2 . w and the kernel's epilogue multiplies the accumulator by
// 2^-COMBO_F16_BSCALE_LOG2 (both exact): the lo pieces of typical weights (|w| ~ 0.01 ... 0.1) would otherwise be fp16 SUBNORMALS with an absolute
// floor of 2^-25 (1.5e-6 relative at |w| = 0.02, as coarse as the bf16 split at |w| = 0.005); range: 2^8 |w| < 65 504
#define COMBO_F16_BSCALE_LOG2 8
// `products` values: 3 = 3 products on bf16 hi / lo pieces, 1 = plain bf16, 19 (16 + 3) = 3 products on fp16 pieces with a WEIGHT image (split
// from 2^8 . w: the epilogue multiplies by 2^-8), 35 (32 + 3) = the same with an ACTIVATION image (split unscaled: the per-frame mask features
// of the mask-logit contraction, whose range is the activations' 65 504, not a weight's 255)
enum { COMBO_PRODUCTS_F16X3 = 19, COMBO_PRODUCTS_F16X3_UNSCALED = 35 };

struct combo_nt3_conv {
  int H, W, Cin;  // input map and channels
  int tap_step;   // split convolution: batch entry b computes taps b * tap_step .. (b + 1) * tap_step - 1 (K = tap_step * Cin); else 0
  int stride;     // 0 / 1: output map = input map; 2: output (ceil(H / 2), ceil(W / 2)) - the GEMM's M counts OUTPUT tokens
  int pad;        // 1: 3x3 taps around the (strided) centre, zero padding; 0: the single tap of a strided 1x1 convolution
};

// C_b[M,N] = A_b[M,K] . Bimg_b[N,K]^T (+ bias) (+ ReLU) (mask_b > 0 ? . : 0), `batch` problems at base + b * stride (elements; the
// mask shares C's pitch and stride; add: C += add before the ReLU, same layout); products: 3 = fp32-accurate split (bf16 pieces), COMBO_PRODUCTS_F16X3 = the same on fp16 pieces (Bimg from the fp16 pre-split), 1 = plain bf16; conv != nullptr: implicit-GEMM 3x3
// convolution (A = NHWC tokens, K = 9 * Cin); force_cfg: 0 auto, 1 wide (256 x 128), 2 mid (128 x 128), 3 skinny (64 x 64), 4 tall (256 x 64).
int combo_nt3_launch(const float* A, long long lda, const float* Bimg, long long ldb, const float* bias, const float* mask, float* C,
                     long long ldc, long long M, int N, int K, int relu, int products, int batch, long long sA, long long sB,
                     long long sC, const combo_nt3_conv* conv, int force_cfg, combo_stream_t stream, const float* add = nullptr);

// split-K form (K slices as the batch entries of one launch into workspace [splits, M, N] + the fixed-order finishing sum with the epilogue):
// csrc/gemm_nt3.hip; products as above (the forward modes pass theirs, every gradient GEMM 3)
int nt3_split_launch(const float* A, long long lda, const float* Bimg, const float* bias, const float* add, const float* mask, float* C,
                     long long ldc, long long M, int N, int K, int relu, int splits, float* workspace, const combo_nt3_conv* conv,
                     combo_stream_t stream, int products = 3);
