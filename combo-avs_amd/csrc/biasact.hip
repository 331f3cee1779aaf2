// Epilogue of the host-PyTorch backbone convolutions (bf16, NHWC): y = relu(y + bias[c] (+ residual)) in ONE pass, and its
// backward dx = dy * (y > 0).  MIOpen runs conv, bias add and ReLU as three kernels (three passes over the activation);
// FrozenBN is folded into (weight, bias) (backbone.py), so this is all that is left between two convolutions.
// Reference semantics: detectron2 BottleneckBlock.forward (conv -> FrozenBN -> relu; out += shortcut; relu) [d2, not in
// /root/reference; parity unpinned, see DESIGN.md section 2].
#include "combo_common.h"

namespace {

typedef unsigned short u16;

__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ u16 f2bf(float f) {  // round to nearest even
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}

// 8 channels (16 bytes) per thread; C % 8 == 0
__global__ void __launch_bounds__(256)
bias_act_kernel(u16* __restrict__ y, const float* __restrict__ bias, const u16* __restrict__ res, long long n8, int C8,
                int relu) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const int c8 = fast_mod(i, C8);
  uint4 v = reinterpret_cast<const uint4*>(y)[i];
  uint4 r = make_uint4(0, 0, 0, 0);
  if (res) r = reinterpret_cast<const uint4*>(res)[i];
  const float4 b0 = reinterpret_cast<const float4*>(bias)[c8 * 2], b1 = reinterpret_cast<const float4*>(bias)[c8 * 2 + 1];
  const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  unsigned vin[4] = {v.x, v.y, v.z, v.w}, rin[4] = {r.x, r.y, r.z, r.w}, out[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float lo = bf2f((u16)(vin[k] & 0xffffu)) + bb[2 * k], hi = bf2f((u16)(vin[k] >> 16)) + bb[2 * k + 1];
    if (res) { lo += bf2f((u16)(rin[k] & 0xffffu)); hi += bf2f((u16)(rin[k] >> 16)); }
    if (relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
    out[k] = (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
  }
  reinterpret_cast<uint4*>(y)[i] = make_uint4(out[0], out[1], out[2], out[3]);
}

__global__ void __launch_bounds__(256)
relu_grad_kernel(const u16* __restrict__ dy, const u16* __restrict__ y, long long n8, u16* __restrict__ dx) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const uint4 g = reinterpret_cast<const uint4*>(dy)[i];
  const uint4 v = reinterpret_cast<const uint4*>(y)[i];
  const unsigned gin[4] = {g.x, g.y, g.z, g.w}, vin[4] = {v.x, v.y, v.z, v.w};
  unsigned out[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // y is a ReLU output: positive <=> sign bit clear and magnitude non-zero
    const unsigned lo_on = ((vin[k] & 0x7fffu) != 0 && (vin[k] & 0x8000u) == 0) ? 0xffffu : 0u;
    const unsigned hi_on = ((vin[k] & 0x7fff0000u) != 0 && (vin[k] & 0x80000000u) == 0) ? 0xffff0000u : 0u;
    out[k] = gin[k] & (lo_on | hi_on);
  }
  reinterpret_cast<uint4*>(dx)[i] = make_uint4(out[0], out[1], out[2], out[3]);
}

// fp32: dx = dy * (y > 0)  (ReLU backward of the head's fused Linear+ReLU layers; one kernel instead of compare + mul)
__global__ void __launch_bounds__(256)
relu_grad_f32_kernel(const float* __restrict__ dy, const float* __restrict__ y, long long n4, float* __restrict__ dx) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 g = reinterpret_cast<const float4*>(dy)[i];
  const float4 v = reinterpret_cast<const float4*>(y)[i];
  reinterpret_cast<float4*>(dx)[i] = make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
}

// fp32: dx = (dy1 + dy2) * (y > 0): the ReLU backward of a bottleneck block's output whose two consumers (next block's first
// convolution, next block's identity / shortcut branch) hand their gradients over separately - one pass instead of autograd's
// accumulation kernel (read 2, write 1) followed by the ReLU gradient (read 2, write 1)
__global__ void __launch_bounds__(256)
relu_grad2_f32_kernel(const float* __restrict__ dy1, const float* __restrict__ dy2, const float* __restrict__ y, long long n4,
                      float* __restrict__ dx) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 a = reinterpret_cast<const float4*>(dy1)[i];
  const float4 b = reinterpret_cast<const float4*>(dy2)[i];
  const float4 v = reinterpret_cast<const float4*>(y)[i];
  reinterpret_cast<float4*>(dx)[i] = make_float4(v.x > 0.f ? a.x + b.x : 0.f, v.y > 0.f ? a.y + b.y : 0.f, v.z > 0.f ? a.z + b.z : 0.f,
                                                 v.w > 0.f ? a.w + b.w : 0.f);
}

// ... and with three consumers (the last block of a ResNet stage: next stage's first convolution, its shortcut branch, the head)
__global__ void __launch_bounds__(256)
relu_grad3_f32_kernel(const float* __restrict__ dy1, const float* __restrict__ dy2, const float* __restrict__ dy3,
                      const float* __restrict__ y, long long n4, float* __restrict__ dx) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 a = reinterpret_cast<const float4*>(dy1)[i];
  const float4 b = reinterpret_cast<const float4*>(dy2)[i];
  const float4 c = reinterpret_cast<const float4*>(dy3)[i];
  const float4 v = reinterpret_cast<const float4*>(y)[i];
  reinterpret_cast<float4*>(dx)[i] = make_float4(v.x > 0.f ? a.x + b.x + c.x : 0.f, v.y > 0.f ? a.y + b.y + c.y : 0.f,
                                                 v.z > 0.f ? a.z + b.z + c.z : 0.f, v.w > 0.f ? a.w + b.w + c.w : 0.f);
}

// Input gradient of a 1x1 stride-2 convolution from its compact form: dst [B, H, W, C] = src [B, ceil(H/2), ceil(W/2), C] at the even
// pixels, zero elsewhere - one write-only pass (the GEMM dY . W over the OUTPUT tokens produced `src`; round 5: the shortcut
// convolutions of res3.0 / res4.0 / res5.0 leave the library's backward-data kernel).  One float4 of dst per thread.
__global__ void __launch_bounds__(256)
expand_stride2_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int C4, long long n4) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const long long tok = fast_div(i, C4);
  const int c4 = (int)(i - tok * C4);
  const long long row = fast_div(tok, W);
  const int x = (int)(tok - row * W);
  const long long b = fast_div(row, H);
  const int y = (int)(row - b * H);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!((x | y) & 1)) {
    const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
    v = reinterpret_cast<const float4*>(src)[((b * Ho + (y >> 1)) * Wo + (x >> 1)) * C4 + c4];
  }
  reinterpret_cast<float4*>(dst)[i] = v;
}

// fp32 variant of bias_act_kernel (the reference's S4 recipe runs the backbones in fp32: SOLVER.AMP.ENABLED False); 4 channels per thread
__global__ void __launch_bounds__(256)
bias_act_f32_kernel(float* __restrict__ y, const float* __restrict__ bias, const float* __restrict__ res, long long n4, int C4, int relu) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 v = reinterpret_cast<const float4*>(y)[i];
  const float4 b = reinterpret_cast<const float4*>(bias)[fast_mod(i, C4)];
  v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  if (res) {
    const float4 r = reinterpret_cast<const float4*>(res)[i];
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  reinterpret_cast<float4*>(y)[i] = v;
}

}  // namespace

extern "C" {

int combo_bias_act_f32(float* y, const float* bias, const float* residual, long long tokens, int C, int relu, combo_stream_t stream) {
  if (!y || !bias || tokens <= 0 || C <= 0 || C % 4 != 0 || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)residual & 15))
    return COMBO_EINVAL;
  const long long n4 = tokens * (C / 4);
  hipLaunchKernelGGL(bias_act_f32_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, bias, residual, n4,
                     C / 4, relu);
  return (int)hipGetLastError();
}

int combo_relu_grad_f32(const float* dy, const float* y, long long n, float* dx, combo_stream_t stream) {
  if (!dy || !y || !dx || n <= 0 || n % 4 != 0 || ((uintptr_t)dy & 15) || ((uintptr_t)y & 15) || ((uintptr_t)dx & 15))
    return COMBO_EINVAL;
  const long long n4 = n / 4;
  hipLaunchKernelGGL(relu_grad_f32_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, y, n4, dx);
  return (int)hipGetLastError();
}

int combo_relu_grad2_f32(const float* dy1, const float* dy2, const float* y, long long n, float* dx, combo_stream_t stream) {
  if (!dy1 || !dy2 || !y || !dx || n <= 0 || n % 4 != 0 || (((uintptr_t)dy1 | (uintptr_t)dy2 | (uintptr_t)y | (uintptr_t)dx) & 15))
    return COMBO_EINVAL;
  const long long n4 = n / 4;
  hipLaunchKernelGGL(relu_grad2_f32_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy1, dy2, y, n4, dx);
  return (int)hipGetLastError();
}

int combo_relu_grad3_f32(const float* dy1, const float* dy2, const float* dy3, const float* y, long long n, float* dx,
                         combo_stream_t stream) {
  if (!dy1 || !dy2 || !dy3 || !y || !dx || n <= 0 || n % 4 != 0 ||
      (((uintptr_t)dy1 | (uintptr_t)dy2 | (uintptr_t)dy3 | (uintptr_t)y | (uintptr_t)dx) & 15))
    return COMBO_EINVAL;
  const long long n4 = n / 4;
  hipLaunchKernelGGL(relu_grad3_f32_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy1, dy2, dy3, y, n4,
                     dx);
  return (int)hipGetLastError();
}

int combo_expand_stride2_f32(const float* src, float* dst, int B, int H, int W, int C, combo_stream_t stream) {
  if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0 || (((uintptr_t)src | (uintptr_t)dst) & 15)) return COMBO_EINVAL;
  const long long n4 = (long long)B * H * W * (C / 4);
  hipLaunchKernelGGL(expand_stride2_f32_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, C / 4, n4);
  return (int)hipGetLastError();
}

int combo_bias_act_bf16(void* y, const float* bias, const void* residual, long long tokens, int C, int relu,
                        combo_stream_t stream) {
  if (!y || !bias || tokens <= 0 || C <= 0 || C % 8 != 0 || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15) ||
      ((uintptr_t)residual & 15))
    return COMBO_EINVAL;
  const long long n8 = tokens * (C / 8);
  hipLaunchKernelGGL(bias_act_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (u16*)y, bias,
                     (const u16*)residual, n8, C / 8, relu);
  return (int)hipGetLastError();
}

int combo_relu_grad_bf16(const void* dy, const void* y, long long n, void* dx, combo_stream_t stream) {
  if (!dy || !y || !dx || n <= 0 || n % 8 != 0 || ((uintptr_t)dy & 15) || ((uintptr_t)y & 15) || ((uintptr_t)dx & 15))
    return COMBO_EINVAL;
  const long long n8 = n / 8;
  hipLaunchKernelGGL(relu_grad_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const u16*)dy, (const u16*)y, n8, (u16*)dx);
  return (int)hipGetLastError();
}

}  // extern "C"
