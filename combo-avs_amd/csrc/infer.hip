// Inference tail (reference: models/maskformer_model.py:393-402 bilinear upsampling of the Q mask-logit maps to the
// input size, and semantic_inference :460-464  semseg[k] = sum_q softmax(cls[q])[k] * sigmoid(mask[q])).
// Fused: one thread per output pixel walks the Q queries, sampling the low-resolution logits on the fly; the
// [Q, H, W] upsampled tensor (20 MB/frame at Q = 100, 224x224) is never materialised.
#include "combo_common.h"

namespace {

constexpr int KMAX = 8;

template <int KM>
__global__ void __launch_bounds__(256)
semantic_inference_kernel(const float* __restrict__ cls_prob /* [F,Q,K] softmax without the no-object column */,
                          const float* __restrict__ masks /* [F,Q,h,w] */, int F, int Q, int K, int h, int w, int H, int W,
                          float* __restrict__ out /* [F,K,H,W] */) {
  const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t >= (long long)F * H * W) return;
  const int X = (int)(t % W), Y = (int)((t / W) % H), f = (int)(t / ((long long)W * H));
  // ATen upsample_bilinear2d, align_corners = False
  float fy = ((float)h / H) * (Y + 0.5f) - 0.5f, fx = ((float)w / W) * (X + 0.5f) - 0.5f;
  fy = fy < 0.f ? 0.f : fy;
  fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int yp = (y0 < h - 1) ? w : 0, xp = (x0 < w - 1) ? 1 : 0;
  const float ly1 = fy - y0, ly0 = 1.f - ly1, lx1 = fx - x0, lx0 = 1.f - lx1;
  float acc[KM];
#pragma unroll
  for (int k = 0; k < KM; ++k) acc[k] = 0.f;
  const float* m = masks + (long long)f * Q * h * w + y0 * w + x0;
  const float* cp = cls_prob + (long long)f * Q * K;
  for (int q = 0; q < Q; ++q, m += h * w, cp += K) {
    const float v = ly0 * (lx0 * m[0] + lx1 * m[xp]) + ly1 * (lx0 * m[yp] + lx1 * m[yp + xp]);
    const float s = 1.f / (1.f + __expf(-v));
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (k < K) acc[k] += cp[k] * s;
  }
#pragma unroll
  for (int k = 0; k < KM; ++k)
    if (k < K) out[(((long long)f * K + k) * H + Y) * W + X] = acc[k];
}

}  // namespace

extern "C" int combo_semantic_inference_f32(const float* cls_prob, const float* masks, int F, int Q, int K, int h, int w,
                                            int H, int W, float* out, combo_stream_t stream) {
  if (!cls_prob || !masks || !out || F <= 0 || Q <= 0 || K <= 0 || K > KMAX || h <= 0 || w <= 0 || H <= 0 || W <= 0)
    return COMBO_EINVAL;  // K > 8 (AVSS, K = 71): the caller walks the classes in groups of 8
  const long long n = (long long)F * H * W;
  hipLaunchKernelGGL((semantic_inference_kernel<KMAX>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, cls_prob, masks, F, Q, K, h, w, H, W, out);
  return (int)hipGetLastError();
}
