/* CPU ORACLE (test infrastructure only) — plain-C restatement of the MSDeformAttn core op.
 *
 * Follows the reference's device code line by line in MEANING, not in form:
 *   forward  : ms_deform_attn_im2col_bilinear + ms_deformable_im2col_gpu_kernel
 *              (models/modeling/pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh:38-89, 242-304)
 *   backward : ms_deform_attn_col2im_bilinear (+ the per-block reduction over channels)
 *              (.cuh:92-164, 306-408)
 * One scalar loop nest per output element, double or float by macro.  Pinned against the golden vectors produced by
 * the reference's ms_deform_attn_core_pytorch (tests/golden/msda_core.npz) in tests/test_oracle_golden.py.
 * Never linked into the product.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define DEFINE_MSDA(T, SUF)                                                                                         \
  void msda_core_forward_##SUF(const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc, const T* aw, \
                               int B, int S, int M, int D, int L, int Lq, int P, T* out) {                         \
    for (long b = 0; b < B; ++b)                                                                                    \
      for (long q = 0; q < Lq; ++q)                                                                                 \
        for (long m = 0; m < M; ++m)                                                                                \
          for (long c = 0; c < D; ++c) {                                                                            \
            T col = 0;                                                                                              \
            const long samp = (b * Lq + q) * M + m;                                                                 \
            for (int l = 0; l < L; ++l) {                                                                           \
              const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                                         \
              const T* v = value + ((long)b * S + lsi[l]) * M * D;                                                  \
              for (int p = 0; p < P; ++p) {                                                                         \
                const long e = (samp * L + l) * P + p;                                                              \
                const T w_im = loc[2 * e] * W - (T)0.5, h_im = loc[2 * e + 1] * H - (T)0.5; /* .cuh:290-291 */      \
                if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) { /* .cuh:294 */                               \
                  const int h0 = (int)floor(h_im), w0 = (int)floor(w_im), h1 = h0 + 1, w1 = w0 + 1;                 \
                  const T lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;                                 \
                  T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                                                 \
                  if (h0 >= 0 && w0 >= 0) v1 = v[((long)h0 * W + w0) * M * D + m * D + c];                          \
                  if (h0 >= 0 && w1 <= W - 1) v2 = v[((long)h0 * W + w1) * M * D + m * D + c];                      \
                  if (h1 <= H - 1 && w0 >= 0) v3 = v[((long)h1 * W + w0) * M * D + m * D + c];                      \
                  if (h1 <= H - 1 && w1 <= W - 1) v4 = v[((long)h1 * W + w1) * M * D + m * D + c];                  \
                  col += (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4) * aw[e]; /* .cuh:87, 297 */    \
                }                                                                                                   \
              }                                                                                                     \
            }                                                                                                       \
            out[samp * D + c] = col;                                                                                \
          }                                                                                                         \
  }                                                                                                                 \
  void msda_core_backward_##SUF(const T* gout, const T* value, const int64_t* shapes, const int64_t* lsi,           \
                                const T* loc, const T* aw, int B, int S, int M, int D, int L, int Lq, int P,        \
                                T* gvalue, T* gloc, T* gaw) {                                                       \
    memset(gvalue, 0, sizeof(T) * (size_t)B * S * M * D);                                                           \
    memset(gloc, 0, sizeof(T) * (size_t)B * Lq * M * L * P * 2);                                                    \
    memset(gaw, 0, sizeof(T) * (size_t)B * Lq * M * L * P);                                                         \
    for (long b = 0; b < B; ++b)                                                                                    \
      for (long q = 0; q < Lq; ++q)                                                                                 \
        for (long m = 0; m < M; ++m) {                                                                              \
          const long samp = (b * Lq + q) * M + m;                                                                   \
          for (int l = 0; l < L; ++l) {                                                                             \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                                           \
            const long base = ((long)b * S + lsi[l]) * M * D;                                                       \
            for (int p = 0; p < P; ++p) {                                                                           \
              const long e = (samp * L + l) * P + p;                                                                \
              const T w_im = loc[2 * e] * W - (T)0.5, h_im = loc[2 * e + 1] * H - (T)0.5;                           \
              if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;                                      \
              const int h0 = (int)floor(h_im), w0 = (int)floor(w_im), h1 = h0 + 1, w1 = w0 + 1;                     \
              const T lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;                                     \
              const T w1_ = hh * hw, w2_ = hh * lw, w3_ = lh * hw, w4_ = lh * lw;                                   \
              for (long c = 0; c < D; ++c) { /* the reference reduces these over the D threads of a block */       \
                const T tg = gout[samp * D + c], tgv = tg * aw[e];                                                  \
                T gh = 0, gw = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                                   \
                if (h0 >= 0 && w0 >= 0) {                                                                           \
                  const long o = base + ((long)h0 * W + w0) * M * D + m * D + c;                                    \
                  v1 = value[o]; gh -= hw * v1; gw -= hh * v1; gvalue[o] += w1_ * tgv;                              \
                }                                                                                                   \
                if (h0 >= 0 && w1 <= W - 1) {                                                                       \
                  const long o = base + ((long)h0 * W + w1) * M * D + m * D + c;                                    \
                  v2 = value[o]; gh -= lw * v2; gw += hh * v2; gvalue[o] += w2_ * tgv;                              \
                }                                                                                                   \
                if (h1 <= H - 1 && w0 >= 0) {                                                                       \
                  const long o = base + ((long)h1 * W + w0) * M * D + m * D + c;                                    \
                  v3 = value[o]; gh += hw * v3; gw -= lh * v3; gvalue[o] += w3_ * tgv;                              \
                }                                                                                                   \
                if (h1 <= H - 1 && w1 <= W - 1) {                                                                   \
                  const long o = base + ((long)h1 * W + w1) * M * D + m * D + c;                                    \
                  v4 = value[o]; gh += lw * v4; gw += lh * v4; gvalue[o] += w4_ * tgv;                              \
                }                                                                                                   \
                gaw[e] += tg * (w1_ * v1 + w2_ * v2 + w3_ * v3 + w4_ * v4); /* .cuh:160-161 */                      \
                gloc[2 * e] += W * gw * tgv;                                 /* .cuh:162 */                         \
                gloc[2 * e + 1] += H * gh * tgv;                             /* .cuh:163 */                         \
              }                                                                                                     \
            }                                                                                                       \
          }                                                                                                         \
        }                                                                                                           \
  }

DEFINE_MSDA(float, f32)
DEFINE_MSDA(double, f64)
