// Residual add + LayerNorm, forward and input gradient, for the post-norm layers of the head:
//   encoder   src = LN(src + attn(src)), src = LN(src + ffn(src))           pixel_decoder/msdeformattn.py:119-134
//   decoder   tgt = LN(tgt + xattn), LN(tgt + selfattn), LN(tgt + ffn(tgt))  transformer_decoder/transformer_decoder.py:99-118, 50-58, 178-182
//   decoder_norm (no residual), applied before each of the 10 prediction heads :494
// ATen runs these as an add kernel (2 reads + 1 write), a LayerNorm kernel (1 read + 1 write) and, backward, a
// layer_norm_grad_input kernel + the accumulation add of the two branches.  Here: ONE pass forward (reads x and r, writes
// z = x + r once - it is what the backward needs - and y), ONE pass backward (dz serves both branches: the caller returns the
// same tensor for x and r).  The parameter gradients stay in the deferred grouped launch (csrc/lngrad.hip).
// Round 3: fan-out.  The LN output of a post-norm layer feeds 2-3 consumers (the next block's value / FFN input, its residual,
// and - plus the position embedding - its query): autograd then sums their gradients with one accumulation kernel per extra
// consumer (read 2, write 1 of the full activation each) before it calls this backward, and the forward needs an `y + pos`
// kernel.  The forward kernel now also writes yp = y + pos (pos broadcast over the frames), and the backward kernel takes up
// to FOUR output gradients (the autograd node hands its consumers aliases of y) and sums them on the fly.
// HBM-bound: 16 B per lane per tensor, one wave per row (C = 64 * VEC channels), statistics by DPP/shuffle inside the wave,
// two-pass variance on the registers (no E[x^2] - E[x]^2 cancellation).
#include "combo_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// VEC consecutive floats of a lane as the widest accesses their count allows (rows are 16-byte aligned: C % 4 == 0 wherever
// VEC % 4 == 0).  Round 5: the scalar form compiled to 4 dword loads / stores per tensor and lane (16-byte stride between
// lanes: four partial-line requests where one full one does) - the 41 160-row encoder launches ran at 2.0 TB/s.
template <int VEC>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&v)[VEC]) {
  if constexpr (VEC % 4 == 0) {
#pragma unroll
    for (int i = 0; i < VEC; i += 4) {
      const float4 t = *reinterpret_cast<const float4*>(p + i);
      v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
    }
  } else if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = p[i];
  }
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&v)[VEC]) {
  if constexpr (VEC % 4 == 0) {
#pragma unroll
    for (int i = 0; i < VEC; i += 4) *reinterpret_cast<float4*>(p + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) p[i] = v[i];
  }
}

template <int VEC>  // channels per lane: C = 64 * VEC, VEC in {1, 2, 4, 5, 8} (64 / 320 = PVTv2's widths)
__global__ void __launch_bounds__(256)
add_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ w, const float* __restrict__ b,
                  float eps, long long rows, float* __restrict__ z, float* __restrict__ y, float* __restrict__ mean,
                  float* __restrict__ rstd, const float* __restrict__ pos, long long pos_rows, float* __restrict__ yp) {
  constexpr int C = 64 * VEC;
  const long long row = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const long long off = row * C + lane * VEC;
  float v[VEC];
  load_vec<VEC>(x + off, v);
  if (r) {
    float rv[VEC];
    load_vec<VEC>(r + off, rv);
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] += rv[i];
  }
  float wv[VEC], bv[VEC], pv[VEC];
  load_vec<VEC>(w + lane * VEC, wv);
  load_vec<VEC>(b + lane * VEC, bv);
  if (yp) load_vec<VEC>(pos + (long long)fast_mod(row, (int)pos_rows) * C + lane * VEC, pv);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) s += v[i];
  const float mu = wave_sum(s) * (1.f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) q += (v[i] - mu) * (v[i] - mu);
  const float rs = rsqrtf(wave_sum(q) * (1.f / C) + eps);
  if (z) store_vec<VEC>(z + off, v);
  float o[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) o[i] = (v[i] - mu) * rs * wv[i] + bv[i];
  store_vec<VEC>(y + off, o);
  if (yp) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] += pv[i];
    store_vec<VEC>(yp + off, o);
  }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dz = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  g = dy * w,  xhat = (z - mean) * rstd
template <int VEC>
__global__ void __launch_bounds__(256)
ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ rstd,
              const float* __restrict__ w, long long rows, float* __restrict__ dz, const float* __restrict__ dy2,
              const float* __restrict__ dy3, const float* __restrict__ dy4, float* __restrict__ dy_sum) {
  constexpr int C = 64 * VEC;
  const long long row = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const long long off = row * C + lane * VEC;
  const float mu = mean[row], rs = rstd[row];
  float g[VEC], xh[VEC], d[VEC], t[VEC], wv[VEC];
  float s1 = 0.f, s2 = 0.f;
  load_vec<VEC>(dy + off, d);
  if (dy2) {
    load_vec<VEC>(dy2 + off, t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) d[i] += t[i];
  }
  if (dy3) {
    load_vec<VEC>(dy3 + off, t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) d[i] += t[i];
  }
  if (dy4) {
    load_vec<VEC>(dy4 + off, t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) d[i] += t[i];
  }
  if (dy_sum) store_vec<VEC>(dy_sum + off, d);  // the summed output gradient, for the deferred parameter-gradient launch
  load_vec<VEC>(z + off, t);
  load_vec<VEC>(w + lane * VEC, wv);
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    g[i] = d[i] * wv[i];
    xh[i] = (t[i] - mu) * rs;
    s1 += g[i];
    s2 += g[i] * xh[i];
  }
  s1 = wave_sum(s1) * (1.f / C);
  s2 = wave_sum(s2) * (1.f / C);
  float o[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) o[i] = rs * (g[i] - s1 - xh[i] * s2);
  store_vec<VEC>(dz + off, o);
}

inline bool c_ok(int C) { return C == 64 || C == 128 || C == 256 || C == 320 || C == 512; }

}  // namespace

extern "C" int combo_add_layernorm_forward_f32(const float* x, const float* r, const float* w, const float* b, float eps, long long rows,
                                               int C, float* z, float* y, float* mean, float* rstd, const float* pos,
                                               long long pos_rows, float* yp, combo_stream_t stream) {
  if (!x || !w || !b || !y || !mean || !rstd || rows <= 0 || !c_ok(C) ||
      (((uintptr_t)x | (uintptr_t)r | (uintptr_t)z | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b | (uintptr_t)pos | (uintptr_t)yp) & 15) ||
      ((pos != nullptr) != (yp != nullptr)) || (pos && pos_rows <= 0))
    return COMBO_EINVAL;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  switch (C / 64) {
    case 1: hipLaunchKernelGGL(add_ln_fwd_kernel<1>, grid, block, 0, (hipStream_t)stream, x, r, w, b, eps, rows, z, y, mean, rstd, pos, pos_rows, yp); break;
    case 2: hipLaunchKernelGGL(add_ln_fwd_kernel<2>, grid, block, 0, (hipStream_t)stream, x, r, w, b, eps, rows, z, y, mean, rstd, pos, pos_rows, yp); break;
    case 4: hipLaunchKernelGGL(add_ln_fwd_kernel<4>, grid, block, 0, (hipStream_t)stream, x, r, w, b, eps, rows, z, y, mean, rstd, pos, pos_rows, yp); break;
    case 5: hipLaunchKernelGGL(add_ln_fwd_kernel<5>, grid, block, 0, (hipStream_t)stream, x, r, w, b, eps, rows, z, y, mean, rstd, pos, pos_rows, yp); break;
    default: hipLaunchKernelGGL(add_ln_fwd_kernel<8>, grid, block, 0, (hipStream_t)stream, x, r, w, b, eps, rows, z, y, mean, rstd, pos, pos_rows, yp); break;
  }
  return (int)hipGetLastError();
}

extern "C" int combo_layernorm_backward_f32(const float* dy, const float* z, const float* mean, const float* rstd, const float* w,
                                            long long rows, int C, float* dz, const float* dy2, const float* dy3, const float* dy4,
                                            float* dy_sum, combo_stream_t stream) {
  if (!dy || !z || !mean || !rstd || !w || !dz || rows <= 0 || !c_ok(C) ||
      (((uintptr_t)dy | (uintptr_t)z | (uintptr_t)dz | (uintptr_t)w | (uintptr_t)dy2 | (uintptr_t)dy3 | (uintptr_t)dy4 |
        (uintptr_t)dy_sum) & 15))
    return COMBO_EINVAL;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  switch (C / 64) {
    case 1: hipLaunchKernelGGL(ln_bwd_kernel<1>, grid, block, 0, (hipStream_t)stream, dy, z, mean, rstd, w, rows, dz, dy2, dy3, dy4, dy_sum); break;
    case 2: hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, block, 0, (hipStream_t)stream, dy, z, mean, rstd, w, rows, dz, dy2, dy3, dy4, dy_sum); break;
    case 4: hipLaunchKernelGGL(ln_bwd_kernel<4>, grid, block, 0, (hipStream_t)stream, dy, z, mean, rstd, w, rows, dz, dy2, dy3, dy4, dy_sum); break;
    case 5: hipLaunchKernelGGL(ln_bwd_kernel<5>, grid, block, 0, (hipStream_t)stream, dy, z, mean, rstd, w, rows, dz, dy2, dy3, dy4, dy_sum); break;
    default: hipLaunchKernelGGL(ln_bwd_kernel<8>, grid, block, 0, (hipStream_t)stream, dy, z, mean, rstd, w, rows, dz, dy2, dy3, dy4, dy_sum); break;
  }
  return (int)hipGetLastError();
}
