// Pre-norm residual step of a PVTv2 block (reference: models/modeling/backbone/pvtv2.py:162-175,
//   x = x + drop_path(attn(norm1(x)));  x = x + drop_path(mlp(norm2(x)))),  as ONE pass per LayerNorm:
//   forward   z = x + s[b] * r        (fp32 residual stream; r = the previous branch's bf16 output, s = the stochastic-depth
//             y = LN(z) -> bf16        multiplier of sample b, 1 when absent)   - the input of the next branch, already in the
//                                                                                 compute dtype
//   backward  d  = LN'(dy + dy2) + dz (dy, dy2 bf16 from the branch's one or two consumers of y, dz fp32 from the later stream)
//             dx = d (fp32),  dr = bf16(s[b] * d),  dy32 = float(dy) for the deferred parameter-gradient launch (csrc/lngrad.hip)
// The host-PyTorch formulation runs, per LayerNorm, a cast (fp32 -> bf16), a stochastic-depth multiply, a mixed-dtype add and
// the LayerNorm forward, and their four counterparts backward: ~1 900 launches and ~20 ms of a 160 ms PVTv2-B5 step at
// 4 clips x 10 frames.  HBM-bound: forward reads 4 + 2 and writes 4 + 2 bytes per element, backward reads 2 + 4 + 4 and
// writes 4 + 2 (+ 4).  One wave per row, C = 64 * VEC, statistics in registers (two-pass variance), as csrc/layernorm.hip.
// Round 5: a lane owns the channels {lane, 64 + lane, ...} (it owned VEC consecutive ones): a wave-wide dword / ushort access is
// then 256 / 128 contiguous bytes.  The consecutive form compiled to VEC scalar accesses per tensor with a VEC-element stride
// between lanes (C = 320: 20 bytes - no vector form exists), i.e. VEC partial-line requests where one full one does; these
// kernels were 10 % of the `pvt_avss_512` step.
#include "combo_common.h"

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ unsigned short f2bf(float v) {
  unsigned u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

struct PreArgs {
  const float* x; const unsigned short* r; const float* scale; const float* w; const float* b;
  float eps; long long rows, rows_per_sample;
  float* z; void* y; int y_bf16; float* mean; float* rstd;
};

template <int VEC>
__global__ void __launch_bounds__(256)
prenorm_fwd_kernel(const PreArgs a) {
  constexpr int C = 64 * VEC;
  const long long row = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int lane = threadIdx.x & 63;
  const long long off = row * C + lane;  // element i of a lane is channel 64 i + lane: every access of a wave is one contiguous run
  float v[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = a.x[off + 64 * i];
  if (a.r) {
    const float s = a.scale ? a.scale[fast_div(row, (int)a.rows_per_sample)] : 1.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] += s * bf2f(a.r[off + 64 * i]);
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) sum += v[i];
  const float mu = wsum(sum) * (1.f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) q += (v[i] - mu) * (v[i] - mu);
  const float rs = rsqrtf(wsum(q) * (1.f / C) + a.eps);
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    if (a.z) a.z[off + 64 * i] = v[i];
    const float o = (v[i] - mu) * rs * a.w[64 * i + lane] + a.b[64 * i + lane];
    if (a.y_bf16) reinterpret_cast<unsigned short*>(a.y)[off + 64 * i] = f2bf(o);
    else reinterpret_cast<float*>(a.y)[off + 64 * i] = o;
  }
  if (lane == 0) { a.mean[row] = mu; a.rstd[row] = rs; }
}

struct PreBwdArgs {
  const void* dy; const void* dy2; int dy_bf16; const float* dz; const float* z; const float* mean; const float* rstd; const float* w;
  const float* scale; long long rows, rows_per_sample;
  float* dx; unsigned short* dr; float* dy32;
};

template <int VEC>
__global__ void __launch_bounds__(256)
prenorm_bwd_kernel(const PreBwdArgs a) {
  constexpr int C = 64 * VEC;
  const long long row = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int lane = threadIdx.x & 63;
  const long long off = row * C + lane;  // element i of a lane is channel 64 i + lane: every access of a wave is one contiguous run
  float d[VEC];
  if (a.dy) {
    const float mu = a.mean[row], rs = a.rstd[row];
    float g[VEC], xh[VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float t = a.dy_bf16 ? bf2f(reinterpret_cast<const unsigned short*>(a.dy)[off + 64 * i]) : reinterpret_cast<const float*>(a.dy)[off + 64 * i];
      if (a.dy2)  // the second consumer of y (fan-out: the autograd node hands out two aliases of y)
        t += a.dy_bf16 ? bf2f(reinterpret_cast<const unsigned short*>(a.dy2)[off + 64 * i]) : reinterpret_cast<const float*>(a.dy2)[off + 64 * i];
      if (a.dy32) a.dy32[off + 64 * i] = t;
      g[i] = t * a.w[64 * i + lane];
      xh[i] = (a.z[off + 64 * i] - mu) * rs;
      s1 += g[i];
      s2 += g[i] * xh[i];
    }
    s1 = wsum(s1) * (1.f / C);
    s2 = wsum(s2) * (1.f / C);
#pragma unroll
    for (int i = 0; i < VEC; ++i) d[i] = rs * (g[i] - s1 - xh[i] * s2);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) d[i] = 0.f;
  }
  if (a.dz) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) d[i] += a.dz[off + 64 * i];
  }
  const float s = (a.dr && a.scale) ? a.scale[fast_div(row, (int)a.rows_per_sample)] : 1.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    a.dx[off + 64 * i] = d[i];
    if (a.dr) a.dr[off + 64 * i] = f2bf(s * d[i]);
  }
}


// Channel bias + LayerNorm on bf16 rows (the key / value path of a spatial-reduction attention block, pvtv2.py:104-108:
// `self.norm(self.sr(x_))`): y = LN(x + xb[c]) -> bf16, backward dx = LN'(dy) -> bf16 (+ fp32 dy for the deferred parameter
// gradients).  The convolution runs without its bias; the bias gradient is the channel sum of dx (csrc/colsum.hip).
struct BiasLnArgs {
  const unsigned short* x; const void* xb; int xb_bf16; const float* w; const float* b; float eps; long long rows;
  unsigned short* y; float* mean; float* rstd;
  const unsigned short* dy; unsigned short* dx; float* dy32; float* z32;
};

template <int VEC, bool BWD>
__global__ void __launch_bounds__(256)
bias_ln_kernel(const BiasLnArgs a) {
  constexpr int C = 64 * VEC;
  const long long row = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int lane = threadIdx.x & 63;
  const long long off = row * C + lane;  // element i of a lane is channel 64 i + lane: every access of a wave is one contiguous run
  float v[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int c = 64 * i + lane;
    const float xb = !a.xb ? 0.f : a.xb_bf16 ? bf2f(reinterpret_cast<const unsigned short*>(a.xb)[c]) : reinterpret_cast<const float*>(a.xb)[c];
    v[i] = bf2f(a.x[off + 64 * i]) + xb;
  }
  if (!BWD) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) sum += v[i];
    const float mu = wsum(sum) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) q += (v[i] - mu) * (v[i] - mu);
    const float rs = rsqrtf(wsum(q) * (1.f / C) + a.eps);
#pragma unroll
    for (int i = 0; i < VEC; ++i) a.y[off + 64 * i] = f2bf((v[i] - mu) * rs * a.w[64 * i + lane] + a.b[64 * i + lane]);
    if (lane == 0) { a.mean[row] = mu; a.rstd[row] = rs; }
  } else {
    const float mu = a.mean[row], rs = a.rstd[row];
    float g[VEC], xh[VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float t = bf2f(a.dy[off + 64 * i]);
      if (a.dy32) { a.dy32[off + 64 * i] = t; a.z32[off + 64 * i] = v[i]; }  // (dy, z) in fp32 for the deferred parameter-gradient launch
      g[i] = t * a.w[64 * i + lane];
      xh[i] = (v[i] - mu) * rs;
      s1 += g[i];
      s2 += g[i] * xh[i];
    }
    s1 = wsum(s1) * (1.f / C);
    s2 = wsum(s2) * (1.f / C);
#pragma unroll
    for (int i = 0; i < VEC; ++i) a.dx[off + 64 * i] = f2bf(rs * (g[i] - s1 - xh[i] * s2));
  }
}

inline bool c_ok(int C) { return C == 64 || C == 128 || C == 256 || C == 320 || C == 512; }

template <typename F>
void by_width(int C, F&& f) {
  switch (C / 64) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 5: f(std::integral_constant<int, 5>{}); break;
    default: f(std::integral_constant<int, 8>{}); break;
  }
}

}  // namespace

extern "C" int combo_prenorm_forward(const float* x, const void* r_bf16, const float* scale, long long rows_per_sample, const float* w,
                                     const float* b, float eps, long long rows, int C, float* z, void* y, int y_bf16, float* mean,
                                     float* rstd, combo_stream_t stream) {
  if (!x || !w || !b || !y || !mean || !rstd || rows <= 0 || !c_ok(C) || (r_bf16 && !z) || (scale && rows_per_sample <= 0) ||
      (((uintptr_t)x | (uintptr_t)z | (uintptr_t)w | (uintptr_t)b) & 15) || (((uintptr_t)r_bf16 | (uintptr_t)y) & 7))
    return COMBO_EINVAL;
  PreArgs a{x, reinterpret_cast<const unsigned short*>(r_bf16), scale, w, b, eps, rows, rows_per_sample > 0 ? rows_per_sample : rows,
            z, y, y_bf16, mean, rstd};
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  by_width(C, [&](auto v) { hipLaunchKernelGGL(prenorm_fwd_kernel<decltype(v)::value>, grid, block, 0, (hipStream_t)stream, a); });
  return (int)hipGetLastError();
}

extern "C" int combo_prenorm_backward(const void* dy, const void* dy2, int dy_bf16, const float* dz, const float* z, const float* mean, const float* rstd,
                                      const float* w, const float* scale, long long rows_per_sample, long long rows, int C, float* dx,
                                      void* dr_bf16, float* dy32, combo_stream_t stream) {
  if ((!dy && !dz) || (dy2 && !dy) || !dx || rows <= 0 || !c_ok(C) || (dy && (!z || !mean || !rstd || !w)) || (scale && rows_per_sample <= 0) ||
      (((uintptr_t)dz | (uintptr_t)z | (uintptr_t)dx | (uintptr_t)w | (uintptr_t)dy32) & 15) || (((uintptr_t)dy | (uintptr_t)dy2 | (uintptr_t)dr_bf16) & 7))
    return COMBO_EINVAL;
  PreBwdArgs a{dy, dy2, dy_bf16, dz, z, mean, rstd, w, scale, rows, rows_per_sample > 0 ? rows_per_sample : rows, dx,
               reinterpret_cast<unsigned short*>(dr_bf16), dy32};
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  by_width(C, [&](auto v) { hipLaunchKernelGGL(prenorm_bwd_kernel<decltype(v)::value>, grid, block, 0, (hipStream_t)stream, a); });
  return (int)hipGetLastError();
}

extern "C" int combo_bias_ln_bf16_forward(const void* x, const void* xb, int xb_bf16, const float* w, const float* b, float eps,
                                          long long rows, int C, void* y, float* mean, float* rstd, combo_stream_t stream) {
  if (!x || !w || !b || !y || !mean || !rstd || rows <= 0 || !c_ok(C) || (((uintptr_t)x | (uintptr_t)y) & 7)) return COMBO_EINVAL;
  BiasLnArgs a{reinterpret_cast<const unsigned short*>(x), xb, xb_bf16, w, b, eps, rows, reinterpret_cast<unsigned short*>(y), mean, rstd,
               nullptr, nullptr, nullptr, nullptr};
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  by_width(C, [&](auto v) { hipLaunchKernelGGL((bias_ln_kernel<decltype(v)::value, false>), grid, block, 0, (hipStream_t)stream, a); });
  return (int)hipGetLastError();
}

extern "C" int combo_bias_ln_bf16_backward(const void* dy, const void* x, const void* xb, int xb_bf16, const float* mean,
                                           const float* rstd, const float* w, long long rows, int C, void* dx, float* dy32,
                                           float* z32, combo_stream_t stream) {
  if (!dy || !x || !mean || !rstd || !w || !dx || rows <= 0 || !c_ok(C) || ((dy32 != nullptr) != (z32 != nullptr)) || (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 7))
    return COMBO_EINVAL;
  BiasLnArgs a{reinterpret_cast<const unsigned short*>(x), xb, xb_bf16, w, nullptr, 0.f, rows, nullptr, const_cast<float*>(mean),
               const_cast<float*>(rstd), reinterpret_cast<const unsigned short*>(dy), reinterpret_cast<unsigned short*>(dx), dy32, z32};
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  by_width(C, [&](auto v) { hipLaunchKernelGGL((bias_ln_kernel<decltype(v)::value, true>), grid, block, 0, (hipStream_t)stream, a); });
  return (int)hipGetLastError();
}
