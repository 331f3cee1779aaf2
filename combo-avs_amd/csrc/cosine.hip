// Frame-to-frame cosine loss on the intermediate mask logits (reference: criterion.py:208-231, 282-286).
// The loss needs, for every intermediate head i and frame t of a clip, |x_t|^2 and x_t . x_{t+1} over the flattened
// Q*HW = 313 600 logits.  `cosine_stats` produces exactly those two reductions in one pass (the tiny scalar math
// c*exp(-c) stays in torch and autograd differentiates it); `cosine_grad` turns the gradients of the reductions back
// into d/dx:  grad x_t = 2 g_nrm[t] x_t + g_dot[t] x_{t+1} + g_dot[t-1] x_{t-1}   (neighbours inside the clip only).
#include "combo_common.h"

namespace {

constexpr int THREADS = 256;

// grid: (chunks, N*BT); x [N*BT, E]
__global__ void __launch_bounds__(THREADS)
cosine_stats_kernel(const float* __restrict__ x, long long E, int n_frame, float* __restrict__ dot, float* __restrict__ nrm) {
  __shared__ float red[2][THREADS / 64];
  const long long row = blockIdx.y;
  const int t = (int)(row % n_frame);
  const bool has_next = t + 1 < n_frame;
  const float4* a = reinterpret_cast<const float4*>(x + row * E);
  const float4* b = reinterpret_cast<const float4*>(x + (row + (has_next ? 1 : 0)) * E);
  const long long n4 = E >> 2;
  float sd = 0.f, sn = 0.f;
  for (long long i = blockIdx.x * (long long)THREADS + threadIdx.x; i < n4; i += (long long)gridDim.x * THREADS) {
    const float4 u = a[i], v = b[i];
    sn += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
    sd += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) { sd += __shfl_xor(sd, s); sn += __shfl_xor(sn, s); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sd; red[1][threadIdx.x >> 6] = sn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float d = 0.f, n = 0.f;
    for (int w = 0; w < THREADS / 64; ++w) { d += red[0][w]; n += red[1][w]; }
    __hip_atomic_fetch_add(nrm + row, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (has_next) __hip_atomic_fetch_add(dot + row, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void __launch_bounds__(THREADS)
cosine_grad_kernel(const float* __restrict__ x, long long E, int n_frame, const float* __restrict__ gdot,
                   const float* __restrict__ gnrm, float* __restrict__ grad, long long perm_inner, long long perm_outer) {
  const long long row = blockIdx.y;
  const int t = (int)(row % n_frame);
  const float cs = 2.f * gnrm[row];
  const float cn = (t + 1 < n_frame) ? gdot[row] : 0.f;
  const float cp = (t > 0) ? gdot[row - 1] : 0.f;
  const float4* xs = reinterpret_cast<const float4*>(x + row * E);
  const float4* xn = reinterpret_cast<const float4*>(x + (row + (t + 1 < n_frame ? 1 : 0)) * E);
  const float4* xp = reinterpret_cast<const float4*>(x + (row - (t > 0 ? 1 : 0)) * E);
  // output row: `row` itself, or - rows = [outer', inner] (head, frame) - the row of the transposed [inner, outer] stack
  const long long orow = perm_outer > 0 ? (row % perm_inner) * perm_outer + row / perm_inner : row;
  float4* g = reinterpret_cast<float4*>(grad + orow * E);
  const long long n4 = E >> 2;
  for (long long i = blockIdx.x * (long long)THREADS + threadIdx.x; i < n4; i += (long long)gridDim.x * THREADS) {
    const float4 s = xs[i], n = xn[i], p = xp[i];
    g[i] = make_float4(cs * s.x + cn * n.x + cp * p.x, cs * s.y + cn * n.y + cp * p.y, cs * s.z + cn * n.z + cp * p.z,
                       cs * s.w + cn * n.w + cp * p.w);
  }
}

}  // namespace

extern "C" {

// x [rows, E] (rows = heads * BT frames, clips of n_frame consecutive rows), dot/nrm [rows] zero-filled by the caller
int combo_cosine_stats_f32(const float* x, long long rows, long long E, int n_frame, float* dot, float* nrm, combo_stream_t stream) {
  if (!x || !dot || !nrm || rows <= 0 || E <= 0 || (E & 3) || n_frame <= 0 || rows % n_frame) return COMBO_EINVAL;
  hipLaunchKernelGGL(cosine_stats_kernel, dim3(16, (unsigned)rows), dim3(THREADS), 0, (hipStream_t)stream, x, E, n_frame, dot, nrm);
  return (int)hipGetLastError();
}

int combo_cosine_grad_f32(const float* x, long long rows, long long E, int n_frame, const float* gdot, const float* gnrm,
                          float* grad, long long perm_inner, long long perm_outer, combo_stream_t stream) {
  if (!x || !gdot || !gnrm || !grad || rows <= 0 || E <= 0 || (E & 3) || n_frame <= 0 || rows % n_frame) return COMBO_EINVAL;
  if (perm_outer > 0 && (perm_inner <= 0 || rows % perm_inner || rows / perm_inner > perm_outer)) return COMBO_EINVAL;
  hipLaunchKernelGGL(cosine_grad_kernel, dim3(16, (unsigned)rows), dim3(THREADS), 0, (hipStream_t)stream, x, E, n_frame, gdot, gnrm,
                     grad, perm_inner, perm_outer);
  return (int)hipGetLastError();
}

}  // extern "C"
