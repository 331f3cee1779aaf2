// Fused full-model gradient clipping + AdamW on a flat fp32 parameter segment
// (reference semantics: train_net.py:205-221 = clip_grad_norm_(all params, 0.01) then torch.optim.AdamW.step).
// One pass: reads g, p, m, v and writes p, m, v (28 B/element) instead of ~10 elementwise passes.
#include "combo_common.h"

namespace {

__global__ void __launch_bounds__(256)
adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
             long long n, const float* __restrict__ clip_coef, float lr, float wd, float b1, float b2, float eps,
             float bc1, float bc2_sqrt) {
  const float clip = clip_coef ? *clip_coef : 1.f;
  const float step = lr / bc1;
  const float decay = 1.f - lr * wd;
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 P = reinterpret_cast<float4*>(p)[i];
    const float4 G = reinterpret_cast<const float4*>(g)[i];
    float4 M = reinterpret_cast<float4*>(m)[i];
    float4 V = reinterpret_cast<float4*>(v)[i];
    float* pp = &P.x; const float* gg = &G.x; float* mm = &M.x; float* vv = &V.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = gg[k] * clip;
      pp[k] *= decay;
      mm[k] = mm[k] * b1 + gk * (1.f - b1);
      vv[k] = vv[k] * b2 + gk * gk * (1.f - b2);
      pp[k] -= step * (mm[k] / (sqrtf(vv[k]) / bc2_sqrt + eps));
    }
    reinterpret_cast<float4*>(p)[i] = P;
    reinterpret_cast<float4*>(m)[i] = M;
    reinterpret_cast<float4*>(v)[i] = V;
  }
  // tail (n % 4)
  const long long base = n4 << 2;
  const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t < n - base) {
    const long long i = base + t;
    const float gk = g[i] * clip;
    float pk = p[i] * decay;
    const float mk = m[i] * b1 + gk * (1.f - b1);
    const float vk = v[i] * b2 + gk * gk * (1.f - b2);
    pk -= step * (mk / (sqrtf(vk) / bc2_sqrt + eps));
    p[i] = pk; m[i] = mk; v[i] = vk;
  }
}

}  // namespace

extern "C" int combo_adamw_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                               const float* clip_coef, float lr, float weight_decay, float beta1, float beta2, float eps,
                               float bias_correction1, float bias_correction2, combo_stream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0) return COMBO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
       reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
    return COMBO_EINVAL;  // 16-byte aligned segments
  long long blocks = ((n >> 2) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, n, clip_coef, lr, weight_decay, beta1, beta2, eps, bias_correction1,
                     sqrtf(bias_correction2));
  return (int)hipGetLastError();
}
