// GroupNorm (+ optional ReLU) on token-major / channels_last fp32 activations [B, HW, C], forward and backward.
// The pixel decoder's 1x1 / 3x3 convolutions produce channels_last maps (the 1x1 ones are token-major GEMMs, the 3x3 is
// faster in NHWC in MIOpen), but ATen's GroupNorm converts to NCHW and back (249 + 323 us at 40 x 256 x 56 x 56 against
// 81 + 161 us on NCHW input) and returns NCHW, which then forces layout copies around every convolution
// (reference: detectron2 Conv2d wrapper = conv -> GroupNorm(32) -> ReLU, msdeformattn.py:215-224, 271-286 [d2]).
//   forward:  stats partials (thread = channel, coalesced rows) -> finalize mean/rstd per (b, group) -> apply (+ReLU)
//   backward: partials of u = sum dy', v = sum dy' * xhat per (b, channel) (dy' = dy masked by the ReLU) -> finalize
//             s1 = sum_c gamma u, s2 = sum_c gamma v per (b, group), dgamma = sum_b v, dbeta = sum_b u -> apply
//             dx = rstd * (dy' gamma - (s1 + xhat s2) / n)
// All sums are accumulated in a fixed order (deterministic).
#include "combo_common.h"

namespace {

constexpr int kTokSlice = 64;  // tokens per partial

// partial[b][slice][c][2] = (sum x, sum x^2) over the tokens of the slice
__global__ void __launch_bounds__(256)
gn_stats_partial_kernel(const float* __restrict__ x, int HW, int C, int slices, float* __restrict__ part) {
  const int b = blockIdx.x / slices, s = blockIdx.x % slices;
  const int t0 = s * kTokSlice, t1 = min(HW, t0 + kTokSlice);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = 0.f, q = 0.f;
    const float* p = x + ((long long)b * HW + t0) * C + c;
    for (int t = t0; t < t1; ++t, p += C) { const float v = *p; a += v; q += v * v; }
    float* o = part + (((long long)b * slices + s) * C + c) * 2;
    o[0] = a; o[1] = q;
  }
}

// one thread per (b, group): mean / rstd
__global__ void __launch_bounds__(64)
gn_stats_final_kernel(const float* __restrict__ part, int B, int HW, int C, int G, int slices, float eps,
                      float* __restrict__ mean, float* __restrict__ rstd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * G) return;
  const int b = i / G, g = i % G, cpg = C / G;
  double a = 0.0, q = 0.0;
  for (int s = 0; s < slices; ++s)
    for (int k = 0; k < cpg; ++k) {
      const float* o = part + (((long long)b * slices + s) * C + g * cpg + k) * 2;
      a += o[0]; q += o[1];
    }
  const double n = (double)HW * cpg;
  const double m = a / n;
  double var = q / n - m * m;
  if (var < 0.0) var = 0.0;
  mean[i] = (float)m;
  rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ void __launch_bounds__(256)
gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                const float* __restrict__ gamma, const float* __restrict__ beta, long long total4, int HW, int C, int G,
                int relu, float* __restrict__ y) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int C4 = C >> 2;
  const int c = fast_mod(i, C4) * 4;
  const long long tok = fast_div(i, C4);
  const int b = (int)fast_div(tok, HW);
  const int cpg = C / G;
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int g = (c + k) / cpg;
    float o = (r[k] - mean[b * G + g]) * rstd[b * G + g] * gamma[c + k] + beta[c + k];
    r[k] = relu ? fmaxf(o, 0.f) : o;
  }
  reinterpret_cast<float4*>(y)[i] = make_float4(r[0], r[1], r[2], r[3]);
}

// partial[b][slice][c][2] = (sum dy', sum dy' * xhat), dy' = dy * (y > 0) if relu
__global__ void __launch_bounds__(256)
gn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                      const float* __restrict__ mean, const float* __restrict__ rstd, int HW, int C, int G, int slices,
                      int relu, float* __restrict__ part) {
  const int b = blockIdx.x / slices, s = blockIdx.x % slices;
  const int t0 = s * kTokSlice, t1 = min(HW, t0 + kTokSlice);
  const int cpg = C / G;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float m = mean[b * G + c / cpg], rs = rstd[b * G + c / cpg];
    float u = 0.f, v = 0.f;
    long long off = ((long long)b * HW + t0) * C + c;
    for (int t = t0; t < t1; ++t, off += C) {
      float g = dy[off];
      if (relu && !(y[off] > 0.f)) g = 0.f;
      u += g;
      v += g * (x[off] - m) * rs;
    }
    float* o = part + (((long long)b * slices + s) * C + c) * 2;
    o[0] = u; o[1] = v;
  }
}

// block = frame b, thread = channel: uv[b][c] = sum over slices; s1/s2 per (b, group) through LDS
__global__ void __launch_bounds__(1024)
gn_bwd_final_kernel(const float* __restrict__ part, const float* __restrict__ gamma, int C, int G, int slices,
                    float* __restrict__ s12 /* [B][G][2] */, float* __restrict__ uv /* [B][C][2] */) {
  extern __shared__ float sh[];  // [C][2]
  const int b = blockIdx.x, cpg = C / G;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float u = 0.f, v = 0.f;
    for (int s = 0; s < slices; ++s) {
      const float* o = part + (((long long)b * slices + s) * C + c) * 2;
      u += o[0]; v += o[1];
    }
    uv[((long long)b * C + c) * 2] = u;
    uv[((long long)b * C + c) * 2 + 1] = v;
    sh[2 * c] = gamma[c] * u;
    sh[2 * c + 1] = gamma[c] * v;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float a = 0.f, q = 0.f;
    for (int k = 0; k < cpg; ++k) { a += sh[2 * (g * cpg + k)]; q += sh[2 * (g * cpg + k) + 1]; }
    s12[(b * G + g) * 2] = a;
    s12[(b * G + g) * 2 + 1] = q;
  }
}

// dgamma[c] = sum_b v[b][c], dbeta[c] = sum_b u[b][c]
__global__ void __launch_bounds__(256)
gn_bwd_param_kernel(const float* __restrict__ uv, int B, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float dg = 0.f, db = 0.f;
  for (int b = 0; b < B; ++b) { db += uv[((long long)b * C + c) * 2]; dg += uv[((long long)b * C + c) * 2 + 1]; }
  dgamma[c] = dg;
  dbeta[c] = db;
}

__global__ void __launch_bounds__(256)
gn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                    const float* __restrict__ s12, long long total4, int HW, int C, int G, int relu, float* __restrict__ dx) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int C4 = C >> 2;
  const int c = fast_mod(i, C4) * 4;
  const long long tok = fast_div(i, C4);
  const int b = (int)fast_div(tok, HW);
  const int cpg = C / G;
  const float inv_n = 1.f / ((float)HW * cpg);
  const float4 gv = reinterpret_cast<const float4*>(dy)[i];
  const float4 xv = reinterpret_cast<const float4*>(x)[i];
  float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
  if (relu) yv = reinterpret_cast<const float4*>(y)[i];
  float g[4] = {gv.x, gv.y, gv.z, gv.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w}, yy[4] = {yv.x, yv.y, yv.z, yv.w}, r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int gi = b * G + (c + k) / cpg;
    const float rs = rstd[gi];
    const float xh = (xx[k] - mean[gi]) * rs;
    const float gg = (relu && !(yy[k] > 0.f)) ? 0.f : g[k];
    r[k] = rs * (gg * gamma[c + k] - (s12[gi * 2] + xh * s12[gi * 2 + 1]) * inv_n);
  }
  reinterpret_cast<float4*>(dx)[i] = make_float4(r[0], r[1], r[2], r[3]);
}

}  // namespace

extern "C" {

int combo_groupnorm_nhwc_slices(int HW) { return (HW + kTokSlice - 1) / kTokSlice; }

int combo_groupnorm_nhwc_forward_f32(const float* x, const float* gamma, const float* beta, int B, int HW, int C, int G,
                                     float eps, int relu, float* part_ws, float* mean, float* rstd, float* y,
                                     combo_stream_t stream) {
  if (!x || !gamma || !beta || !part_ws || !mean || !rstd || !y || B <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G != 0 ||
      C % 4 != 0 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15))
    return COMBO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int slices = combo_groupnorm_nhwc_slices(HW);
  hipLaunchKernelGGL(gn_stats_partial_kernel, dim3(B * slices), dim3(256), 0, st, x, HW, C, slices, part_ws);
  hipLaunchKernelGGL(gn_stats_final_kernel, dim3((B * G + 63) / 64), dim3(64), 0, st, part_ws, B, HW, C, G, slices, eps, mean, rstd);
  const long long total4 = (long long)B * HW * (C / 4);
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, x, mean, rstd, gamma, beta,
                     total4, HW, C, G, relu, y);
  return (int)hipGetLastError();
}

int combo_groupnorm_nhwc_backward_f32(const float* dy, const float* x, const float* y, const float* mean, const float* rstd,
                                      const float* gamma, int B, int HW, int C, int G, int relu, float* part_ws, float* s12_ws,
                                      float* dx, float* dgamma, float* dbeta, combo_stream_t stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !part_ws || !s12_ws || !dx || !dgamma || !dbeta || (relu && !y) || B <= 0 ||
      HW <= 0 || C <= 0 || C > 1024 || G <= 0 || C % G != 0 || C % 4 != 0 || ((uintptr_t)dy & 15) || ((uintptr_t)x & 15) ||
      ((uintptr_t)dx & 15) || (relu && ((uintptr_t)y & 15)))
    return COMBO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int slices = combo_groupnorm_nhwc_slices(HW);
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(B * slices), dim3(256), 0, st, dy, x, y, mean, rstd, HW, C, G, slices, relu,
                     part_ws);
  const int threads = (C + 63) / 64 * 64;
  float* uv_ws = s12_ws + (long long)B * G * 2;  // s12_ws holds [B][G][2] followed by [B][C][2]
  hipLaunchKernelGGL(gn_bwd_final_kernel, dim3(B), dim3(threads), (size_t)C * 2 * sizeof(float), st, part_ws, gamma, C, G,
                     slices, s12_ws, uv_ws);
  hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((C + 255) / 256), dim3(256), 0, st, uv_ws, B, C, dgamma, dbeta);
  const long long total4 = (long long)B * HW * (C / 4);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, dy, x, y, mean, rstd,
                     gamma, s12_ws, total4, HW, C, G, relu, dx);
  return (int)hipGetLastError();
}

}  // extern "C"
