// Bilinear 2x upsampling (align_corners = False) of channels_last fp32 maps, forward and backward: the FPN top-down step
// of the pixel decoder (msdeformattn.py:349-350: F.interpolate(out[-1], size=cur_fpn.shape[-2:], mode="bilinear")).
// ATen's backward scatters with atomics (391 us at 40 x 256 x 28x28 -> 56x56); here the backward is a gather: every input
// pixel revisits the <= 4 x 4 output pixels that can tap it and re-derives their taps (so border clamping is exact).
#include "combo_common.h"

namespace {

// PyTorch area_pixel_compute_source_index (align_corners = False, scale 0.5): src = max((o + 0.5) * 0.5 - 0.5, 0)
__device__ __forceinline__ void taps(int o, int n_in, int& i0, int& i1, float& l1) {
  const float src = fmaxf((o + 0.5f) * 0.5f - 0.5f, 0.f);
  i0 = (int)src;
  i1 = min(i0 + 1, n_in - 1);
  l1 = src - (float)i0;
}

__global__ void __launch_bounds__(256)
up2_fwd_kernel(const float* __restrict__ x, long long x_bstride4, int B, int H, int W, int C4, float* __restrict__ y,
               const float* __restrict__ add) {  // add (nullable): the lateral branch of the FPN step, y = add + up(x)
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int OH = 2 * H, OW = 2 * W;
  if (i >= (long long)B * OH * OW * C4) return;
  const int c = fast_mod(i, C4);
  long long p = fast_div(i, C4);
  const int ox = fast_mod(p, OW); p = fast_div(p, OW);
  const int oy = fast_mod(p, OH);
  const int b = (int)fast_div(p, OH);
  int y0, y1, x0, x1; float ly, lx;
  taps(oy, H, y0, y1, ly); taps(ox, W, x0, x1, lx);
  const float4* xb = reinterpret_cast<const float4*>(x) + (long long)b * x_bstride4 + c;
  const float4 v00 = xb[((long long)y0 * W + x0) * C4], v01 = xb[((long long)y0 * W + x1) * C4];
  const float4 v10 = xb[((long long)y1 * W + x0) * C4], v11 = xb[((long long)y1 * W + x1) * C4];
  const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
  float4 r = make_float4(w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x,
                         w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y,
                         w00 * v00.z + w01 * v01.z + w10 * v10.z + w11 * v11.z,
                         w00 * v00.w + w01 * v01.w + w10 * v10.w + w11 * v11.w);
  if (add) {  // the interpolated value is rounded to fp32 first, then added: bitwise `add + F.interpolate(x)`
    const float4 a = reinterpret_cast<const float4*>(add)[i];
    r.x = a.x + r.x; r.y = a.y + r.y; r.z = a.z + r.z; r.w = a.w + r.w;
  }
  reinterpret_cast<float4*>(y)[i] = r;
}

__global__ void __launch_bounds__(256)
up2_bwd_kernel(const float* __restrict__ dy, int B, int H, int W, int C4, float* __restrict__ dx, long long dx_bstride4) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)B * H * W * C4) return;
  const int c = fast_mod(i, C4);
  long long p = fast_div(i, C4);
  const int ix = fast_mod(p, W); p = fast_div(p, W);
  const int iy = fast_mod(p, H);
  const int b = (int)fast_div(p, H);
  const int OH = 2 * H, OW = 2 * W;
  const float4* gb = reinterpret_cast<const float4*>(dy) + (long long)b * OH * OW * C4 + c;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int oy = max(0, 2 * iy - 1); oy <= min(OH - 1, 2 * iy + 2); ++oy) {
    int y0, y1; float ly;
    taps(oy, H, y0, y1, ly);
    const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
    if (wy == 0.f) continue;
    for (int ox = max(0, 2 * ix - 1); ox <= min(OW - 1, 2 * ix + 2); ++ox) {
      int x0, x1; float lx;
      taps(ox, W, x0, x1, lx);
      const float wgt = wy * ((x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f));
      if (wgt == 0.f) continue;
      const float4 g = gb[((long long)oy * OW + ox) * C4];
      acc.x += wgt * g.x; acc.y += wgt * g.y; acc.z += wgt * g.z; acc.w += wgt * g.w;
    }
  }
  reinterpret_cast<float4*>(dx)[(long long)b * dx_bstride4 + ((long long)iy * W + ix) * C4 + c] = acc;
}

}  // namespace

extern "C" {

int combo_upsample2x_bilinear_add_nhwc_f32(const float* x, long long x_batch_stride, const float* add, int B, int H, int W, int C,
                                           float* y, combo_stream_t stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0 || x_batch_stride % 4 != 0 || ((uintptr_t)x & 15) ||
      ((uintptr_t)y & 15) || ((uintptr_t)add & 15))
    return COMBO_EINVAL;
  const long long n = (long long)B * 4 * H * W * (C / 4);
  hipLaunchKernelGGL(up2_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x_batch_stride / 4, B, H, W, C / 4, y,
                     add);
  return (int)hipGetLastError();
}

int combo_upsample2x_bilinear_nhwc_f32(const float* x, long long x_batch_stride, int B, int H, int W, int C, float* y,
                                       combo_stream_t stream) {
  return combo_upsample2x_bilinear_add_nhwc_f32(x, x_batch_stride, nullptr, B, H, W, C, y, stream);
}

int combo_upsample2x_bilinear_nhwc_backward_f32(const float* dy, int B, int H, int W, int C, float* dx,
                                                long long dx_batch_stride, combo_stream_t stream) {
  if (!dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0 || dx_batch_stride % 4 != 0 || ((uintptr_t)dy & 15) ||
      ((uintptr_t)dx & 15))
    return COMBO_EINVAL;
  const long long n = (long long)B * H * W * (C / 4);
  hipLaunchKernelGGL(up2_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, B, H, W, C / 4, dx, dx_batch_stride / 4);
  return (int)hipGetLastError();
}

}  // extern "C"
