// Prologue of the MSDeformAttn core (ops/modules/ms_deform_attn.py:101-118): from ONE projection row
//   proj[token] = [ sampling offsets (heads x L x P x 2) | attention logits (heads x L*P) ]
// produce sampling_locations = reference + offset / (W_l, H_l) and attention_weights = softmax over the L*P logits of a
// head, and the backward of both into one d_proj row.  The reference runs two Linear layers, a view, a division, a
// broadcast add and a softmax (5 small kernels writing 0.8 + 0.4 MB per frame that the core immediately re-reads);
// with the two projections merged into one GEMM (ops/linear.py::linear_cat) this kernel is everything in between.
// One thread per (token, head).
#include "combo_common.h"

namespace {

constexpr int kMaxLPp = 16;

// Thread roles (one launch): the first tokens * OC4 threads each own one float4 of the offsets block (a pure
// element-wise map: loc has the same (head, level, point, xy) order as the offsets part of the projection row, only the
// row pitch changes), the remaining tokens * M threads each own the L*P logits of one (token, head) for the softmax.
// Consecutive lanes touch consecutive 16-byte chunks, so both parts are coalesced.
__global__ void __launch_bounds__(256)
msda_prep_fwd_kernel(const float* __restrict__ proj, const float* __restrict__ ref, const float* __restrict__ norm,
                     long long tokens, int Lq, int M, int L, int P, int ref_batch_stride, float* __restrict__ loc,
                     float* __restrict__ attn) {
  const int LP = L * P, OC = M * LP * 2, OC4 = OC >> 2, row = M * LP * 3;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long n_loc = tokens * OC4;
  if (i < n_loc) {
    const long long tok = i / OC4;
    const int c4 = (int)(i - tok * OC4);
    const int q = (int)(tok % Lq);
    const long long b = tok / Lq;
    const float4 o = *reinterpret_cast<const float4*>(proj + tok * row + c4 * 4);
    // elements c4*4 .. +3 = two (x, y) pairs of points j0, j0 + 1 within head (c4*4) / (LP*2)
    const int within = (c4 * 4) % (LP * 2);
    const int j0 = within >> 1;
    const int l0 = j0 / P, l1 = (j0 + 1) / P;
    const float* rf = ref + b * ref_batch_stride + (long long)q * L * 2;
    float4 r;
    r.x = rf[2 * l0] + o.x / norm[2 * l0];
    r.y = rf[2 * l0 + 1] + o.y / norm[2 * l0 + 1];
    r.z = rf[2 * l1] + o.z / norm[2 * l1];
    r.w = rf[2 * l1 + 1] + o.w / norm[2 * l1 + 1];
    *reinterpret_cast<float4*>(loc + tok * OC + c4 * 4) = r;
    return;
  }
  const long long k = i - n_loc;
  if (k >= tokens * M) return;
  const long long tok = k / M;
  const int h = (int)(k - tok * M);
  const float* pl = proj + tok * row + OC + h * LP;
  float* at = attn + (tok * M + h) * LP;
  float lg[kMaxLPp];
  float mx = -3.0e38f;
  // the L*P logits of a (token, head) are 16-byte aligned when L*P % 4 == 0 (every shipped config: 12): float4 accesses,
  // a quarter of the vector-memory instructions of the scalar walk
  const bool vec = (LP & 3) == 0;
  if (vec) {
#pragma unroll
    for (int j = 0; j < kMaxLPp; j += 4)
      if (j < LP) {
        const float4 t = *reinterpret_cast<const float4*>(pl + j);
        lg[j] = t.x; lg[j + 1] = t.y; lg[j + 2] = t.z; lg[j + 3] = t.w;
      }
  } else {
#pragma unroll
    for (int j = 0; j < kMaxLPp; ++j)
      if (j < LP) lg[j] = pl[j];
  }
#pragma unroll
  for (int j = 0; j < kMaxLPp; ++j)
    if (j < LP) mx = fmaxf(mx, lg[j]);
  float z = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxLPp; ++j)
    if (j < LP) { lg[j] = expf(lg[j] - mx); z += lg[j]; }
  const float iz = 1.f / z;
  if (vec) {
#pragma unroll
    for (int j = 0; j < kMaxLPp; j += 4)
      if (j < LP) *reinterpret_cast<float4*>(at + j) = make_float4(lg[j] * iz, lg[j + 1] * iz, lg[j + 2] * iz, lg[j + 3] * iz);
  } else {
#pragma unroll
    for (int j = 0; j < kMaxLPp; ++j)
      if (j < LP) at[j] = lg[j] * iz;
  }
}

__global__ void __launch_bounds__(256)
msda_prep_bwd_kernel(const float* __restrict__ dloc, const float* __restrict__ dattn, const float* __restrict__ attn,
                     const float* __restrict__ norm, long long tokens, int M, int L, int P, float* __restrict__ dproj) {
  const int LP = L * P, OC = M * LP * 2, OC4 = OC >> 2, row = M * LP * 3;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long n_loc = tokens * OC4;
  if (i < n_loc) {
    const long long tok = i / OC4;
    const int c4 = (int)(i - tok * OC4);
    const float4 g = *reinterpret_cast<const float4*>(dloc + tok * OC + c4 * 4);
    const int j0 = ((c4 * 4) % (LP * 2)) >> 1;
    const int l0 = j0 / P, l1 = (j0 + 1) / P;
    *reinterpret_cast<float4*>(dproj + tok * row + c4 * 4) =
        make_float4(g.x / norm[2 * l0], g.y / norm[2 * l0 + 1], g.z / norm[2 * l1], g.w / norm[2 * l1 + 1]);
    return;
  }
  const long long k = i - n_loc;
  if (k >= tokens * M) return;
  const long long tok = k / M;
  const int h = (int)(k - tok * M);
  const float* da = dattn + (tok * M + h) * LP;
  const float* at = attn + (tok * M + h) * LP;
  float* pl = dproj + tok * row + OC + h * LP;
  float a[kMaxLPp], g[kMaxLPp];
  float dot = 0.f;
  const bool vec = (LP & 3) == 0;  // float4 accesses, see the forward kernel
  if (vec) {
#pragma unroll
    for (int j = 0; j < kMaxLPp; j += 4)
      if (j < LP) {
        const float4 ta = *reinterpret_cast<const float4*>(at + j), tg = *reinterpret_cast<const float4*>(da + j);
        a[j] = ta.x; a[j + 1] = ta.y; a[j + 2] = ta.z; a[j + 3] = ta.w;
        g[j] = tg.x; g[j + 1] = tg.y; g[j + 2] = tg.z; g[j + 3] = tg.w;
      }
  } else {
#pragma unroll
    for (int j = 0; j < kMaxLPp; ++j)
      if (j < LP) { a[j] = at[j]; g[j] = da[j]; }
  }
#pragma unroll
  for (int j = 0; j < kMaxLPp; ++j)
    if (j < LP) dot += a[j] * g[j];
  if (vec) {
#pragma unroll
    for (int j = 0; j < kMaxLPp; j += 4)
      if (j < LP)
        *reinterpret_cast<float4*>(pl + j) = make_float4(a[j] * (g[j] - dot), a[j + 1] * (g[j + 1] - dot), a[j + 2] * (g[j + 2] - dot),
                                                         a[j + 3] * (g[j + 3] - dot));
  } else {
#pragma unroll
    for (int j = 0; j < kMaxLPp; ++j)
      if (j < LP) pl[j] = a[j] * (g[j] - dot);
  }
}

}  // namespace

extern "C" {

int combo_msda_prep_forward_f32(const float* proj, const float* ref, const float* normalizer, long long tokens, int Lq,
                                int M, int L, int P, int ref_batch_stride, float* loc, float* attn, combo_stream_t stream) {
  if (!proj || !ref || !normalizer || !loc || !attn || tokens <= 0 || Lq <= 0 || M <= 0 || L <= 0 || P <= 0 ||
      L * P > kMaxLPp || tokens % Lq != 0 || (M * L * P) % 4 != 0 || ((uintptr_t)proj & 15) || ((uintptr_t)loc & 15) ||
      ((uintptr_t)attn & 15))
    return COMBO_EINVAL;
  const long long n = tokens * (M * L * P * 2 / 4) + tokens * M;
  hipLaunchKernelGGL(msda_prep_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, proj, ref,
                     normalizer, tokens, Lq, M, L, P, ref_batch_stride, loc, attn);
  return (int)hipGetLastError();
}

int combo_msda_prep_backward_f32(const float* dloc, const float* dattn, const float* attn, const float* normalizer,
                                 long long tokens, int M, int L, int P, float* dproj, combo_stream_t stream) {
  if (!dloc || !dattn || !attn || !normalizer || !dproj || tokens <= 0 || M <= 0 || L <= 0 || P <= 0 || L * P > kMaxLPp ||
      (M * L * P) % 4 != 0 || ((uintptr_t)dloc & 15) || ((uintptr_t)dproj & 15) || ((uintptr_t)dattn & 15) ||
      ((uintptr_t)attn & 15))
    return COMBO_EINVAL;
  const long long n = tokens * (M * L * P * 2 / 4) + tokens * M;
  hipLaunchKernelGGL(msda_prep_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dloc, dattn,
                     attn, normalizer, tokens, M, L, P, dproj);
  return (int)hipGetLastError();
}

}  // extern "C"
