// Depth-wise 3x3 convolution on token-major (NHWC) bf16 activations: the DWConv inside every PVTv2 MLP
// (models/modeling/backbone/pvtv2.py:377-388: Conv2d(dim, dim, 3, 1, 1, groups=dim) on the [B,H,W,C] token grid).
// MIOpen has no tuned depth-wise bf16 NHWC solver on gfx950: it runs naive_conv_* kernels for forward / backward-data
// and a grouped CK kernel for the weight gradient (2.4 ms per call at 125 440 x 256: 158 ms of a 232 ms PVTv2-B5
// forward+backward).  The op is pure bandwidth (9 MACs per element), so: one thread = one token x 8 channels (16-byte
// bf16 vectors, fp32 accumulate), neighbours come from L1/L2; the weight gradient accumulates 9 taps (+ the bias
// gradient) x 8 channels in registers over a slice of tokens and writes one partial per slice, which the split-K
// reduce kernel (gemm_tn.hip) sums.  Weights stay fp32 (tap-major [9][C]): no per-step cast kernels.
#include "combo_common.h"

namespace {

typedef unsigned short u16;

__device__ __forceinline__ float bf2f(unsigned v16) { return __uint_as_float(v16 << 16); }
__device__ __forceinline__ unsigned f2bf(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40;
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
  f[0] = bf2f(v.x & 0xffffu); f[1] = bf2f(v.x >> 16); f[2] = bf2f(v.y & 0xffffu); f[3] = bf2f(v.y >> 16);
  f[4] = bf2f(v.z & 0xffffu); f[5] = bf2f(v.z >> 16); f[6] = bf2f(v.w & 0xffffu); f[7] = bf2f(v.w >> 16);
}

// y[p,c] = bias[c] + sum_tap wT[tap or 8-tap][c] * x[p + off(tap), c]   (zero padding; flip = 1 gives backward-data)
__global__ void __launch_bounds__(256)
dwconv3x3_kernel(const u16* __restrict__ x, const float* __restrict__ wT, const float* __restrict__ bias, int B, int H,
                 int W, int C8, int flip, u16* __restrict__ y) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long total = (long long)B * H * W * C8;
  if (i >= total) return;
  const int c8 = (int)(i % C8);
  const long long p = i / C8;
  const int w = (int)(p % W), h = (int)((p / W) % H);
  const int C = C8 * 8;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = bias ? bias[c8 * 8 + k] : 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int dy = t / 3 - 1, dx = t % 3 - 1;
    const int hh = h + dy, ww = w + dx;
    if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
    const uint4 v = *reinterpret_cast<const uint4*>(x + ((p + dy * W + dx) * C8 + c8) * 8);
    float xv[8];
    unpack8(v, xv);
    const float* wp = wT + (flip ? 8 - t : t) * C + c8 * 8;
    const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
    acc[0] += w0.x * xv[0]; acc[1] += w0.y * xv[1]; acc[2] += w0.z * xv[2]; acc[3] += w0.w * xv[3];
    acc[4] += w1.x * xv[4]; acc[5] += w1.y * xv[5]; acc[6] += w1.z * xv[6]; acc[7] += w1.w * xv[7];
  }
  uint4 o;
  o.x = f2bf(acc[0]) | (f2bf(acc[1]) << 16); o.y = f2bf(acc[2]) | (f2bf(acc[3]) << 16);
  o.z = f2bf(acc[4]) | (f2bf(acc[5]) << 16); o.w = f2bf(acc[6]) | (f2bf(acc[7]) << 16);
  *reinterpret_cast<uint4*>(y + i * 8) = o;
}


// v3 of the forward / backward-data kernel (round 3): v1 (one output token x 8 channels per thread) executes ~700 instructions
// per 16 bytes written - nine bounds-checked taps with their own 64-bit address arithmetic and 18 weight loads - and runs at
// 1.2 TB/s of (one read + one write) at every size: instruction-bound, not memory-bound (the HBM counters show exactly one
// read and one write of the activation).  Here a thread walks a vertical strip of R output rows at one (w, 8 channels): the
// 72 weights are loaded once, the 3 x 3 window slides down (3 new loads per output instead of 9), column validity is a
// per-thread constant and a row outside the map is a row of zeros.
template <int R>
__global__ void __launch_bounds__(256)
dwconv3x3_strip_kernel(const u16* __restrict__ x, const float* __restrict__ wT, const float* __restrict__ bias, int B, int H,
                       int W, int C8, int flip, u16* __restrict__ y) {
  const int HS = (H + R - 1) / R;
  const long long id = xcd_contiguous(blockIdx.x, gridDim.x) * (long long)blockDim.x + threadIdx.x;
  if (id >= (long long)B * HS * W * C8) return;
  // (b, strip, w, c8), c8 fastest: a wave covers consecutive channels / neighbouring columns = contiguous memory
  unsigned t = (unsigned)id;
  const int c8 = t % (unsigned)C8; t /= (unsigned)C8;
  const int w = t % (unsigned)W; t /= (unsigned)W;
  const int hs = t % (unsigned)HS;
  const int b = t / (unsigned)HS;
  const int h0 = hs * R, C = C8 * 8;
  float wt[9][8];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const float* wp = wT + (flip ? 8 - k : k) * C + c8 * 8;
    const float4 a = *reinterpret_cast<const float4*>(wp), c = *reinterpret_cast<const float4*>(wp + 4);
    wt[k][0] = a.x; wt[k][1] = a.y; wt[k][2] = a.z; wt[k][3] = a.w; wt[k][4] = c.x; wt[k][5] = c.y; wt[k][6] = c.z; wt[k][7] = c.w;
  }
  float bs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bs[k] = bias ? bias[c8 * 8 + k] : 0.f;
  const bool lok = w > 0, rok = w < W - 1;
  const long long rs = (long long)W * C8 * 8;                               // elements per image row
  const u16* xc = x + (((long long)b * H * W + w) * C8 + c8) * 8;           // token (b, 0, w)
  u16* yc = y + (((long long)b * H * W + w) * C8 + c8) * 8;
  const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
  uint4 win[3][3];  // [row of the window][column w - 1, w, w + 1], packed bf16
  auto load_row = [&](int hh, uint4 (&row)[3]) {
    if (hh < 0 || hh >= H) { row[0] = row[1] = row[2] = zero; return; }
    const u16* p = xc + hh * rs;
    row[0] = lok ? *reinterpret_cast<const uint4*>(p - C8 * 8) : zero;
    row[1] = *reinterpret_cast<const uint4*>(p);
    row[2] = rok ? *reinterpret_cast<const uint4*>(p + C8 * 8) : zero;
  };
  load_row(h0 - 1, win[0]);
  load_row(h0, win[1]);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int h = h0 + r;
    if (h >= H) break;
    load_row(h + 1, win[2]);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = bs[k];
#pragma unroll
    for (int iy = 0; iy < 3; ++iy)
#pragma unroll
      for (int ix = 0; ix < 3; ++ix) {
        float xv[8];
        unpack8(win[iy][ix], xv);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += wt[iy * 3 + ix][k] * xv[k];
      }
    uint4 o;
    o.x = f2bf(acc[0]) | (f2bf(acc[1]) << 16); o.y = f2bf(acc[2]) | (f2bf(acc[3]) << 16);
    o.z = f2bf(acc[4]) | (f2bf(acc[5]) << 16); o.w = f2bf(acc[6]) | (f2bf(acc[7]) << 16);
    *reinterpret_cast<uint4*>(yc + h * rs) = o;
#pragma unroll
    for (int ix = 0; ix < 3; ++ix) { win[0][ix] = win[1][ix]; win[1][ix] = win[2][ix]; }
  }
}

// partial[s][tap][c] = sum over the tokens of slice s of dy[p,c] * x[p + off(tap), c];  partial[s][9][c] = sum dy[p,c]
__global__ void __launch_bounds__(256)
dwconv3x3_wgrad_kernel(const u16* __restrict__ x, const u16* __restrict__ dy, int B, int H, int W, int C8, int slices,
                       int tok_per_slice, float* __restrict__ partial) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)slices * C8) return;
  const int c8 = (int)(i % C8), s = (int)(i / C8);
  const long long tokens = (long long)B * H * W;
  const long long p0 = (long long)s * tok_per_slice, p1 = min(tokens, p0 + tok_per_slice);
  float acc[10][8];
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[t][k] = 0.f;
  int w = (int)(p0 % W), h = (int)((p0 / W) % H);
  for (long long p = p0; p < p1; ++p) {
    float g[8];
    unpack8(*reinterpret_cast<const uint4*>(dy + (p * C8 + c8) * 8), g);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[9][k] += g[k];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ddy = t / 3 - 1, ddx = t % 3 - 1;
      const int hh = h + ddy, ww = w + ddx;
      if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
      float xv[8];
      unpack8(*reinterpret_cast<const uint4*>(x + ((p + ddy * W + ddx) * C8 + c8) * 8), xv);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[t][k] += g[k] * xv[k];
    }
    if (++w == W) { w = 0; if (++h == H) h = 0; }
  }
  const int C = C8 * 8;
  float* o = partial + (long long)s * 10 * C + c8 * 8;
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    *reinterpret_cast<float4*>(o + t * C) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    *reinterpret_cast<float4*>(o + t * C + 4) = make_float4(acc[t][4], acc[t][5], acc[t][6], acc[t][7]);
  }
}


// v2 of the weight gradient (round 3): v1 gives one thread a whole slice of tokens to walk serially - ~64 k threads, one wave
// per SIMD, nothing to hide the load latency behind (76 us per launch on average against 10 - 30 us of HBM time).  Here a
// workgroup = 32 channel groups x 32 token lanes: token lane tl takes tokens p0 + tl, p0 + tl + 32, ... of the slice, the 32
// lanes are summed at the end (lane ^ 32 by shuffle, the 16 waves through LDS, one tap at a time) and ONE partial per
// workgroup goes out - 32 x fewer partials per thread than v1 at the same slice count.
__global__ void __launch_bounds__(1024)
dwconv3x3_wgrad2_kernel(const u16* __restrict__ x, const u16* __restrict__ dy, int B, int H, int W, int C8, int slices,
                        int tok_per_slice, float* __restrict__ partial) {
  __shared__ float red[16][32][8];
  const int c8l = threadIdx.x & 31, tl = threadIdx.x >> 5, wave = threadIdx.x >> 6;
  const int c8 = blockIdx.x * 32 + c8l, s = blockIdx.y;
  const long long tokens = (long long)B * H * W;
  const long long p0 = (long long)s * tok_per_slice, p1 = min(tokens, p0 + tok_per_slice);
  float acc[10][8];
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[t][k] = 0.f;
  // (w, h) of the token, carried along instead of two integer divisions per token
  int w = (int)((p0 + tl) % W), h = (int)(((p0 + tl) / W) % H);
  for (long long p = p0 + tl; p < p1; p += 32) {
    float g[8];
    unpack8(*reinterpret_cast<const uint4*>(dy + (p * C8 + c8) * 8), g);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[9][k] += g[k];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ddy = t / 3 - 1, ddx = t % 3 - 1;
      const int hh = h + ddy, ww = w + ddx;
      if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
      float xv[8];
      unpack8(*reinterpret_cast<const uint4*>(x + ((p + ddy * W + ddx) * C8 + c8) * 8), xv);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[t][k] += g[k] * xv[k];
    }
    w += 32;
    while (w >= W) { w -= W; if (++h == H) h = 0; }
  }
  const int C = C8 * 8;
  float* o = partial + (long long)s * 10 * C;
#pragma unroll
  for (int t = 0; t < 10; ++t) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[t][k] += __shfl_xor(acc[t][k], 32, 64);  // the wave's two token lanes
    if ((threadIdx.x & 32) == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) red[wave][c8l][k] = acc[t][k];
    }
    __syncthreads();
    if (threadIdx.x < 256) {  // (channel group, k): fixed order over the 16 waves
      const int cl = threadIdx.x >> 3, k = threadIdx.x & 7;
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) v += red[q][cl][k];
      o[t * C + (blockIdx.x * 32 + cl) * 8 + k] = v;
    }
    __syncthreads();
  }
}

// v3 of the weight gradient (round 5): v2's token lane re-reads the 3 x 3 neighbourhood of every token - 10 16-byte loads per
// token and channel group, 9 of them through L1 / L2 again (104 launches, 5.7 ms of a 108 ms `pvt_ms3_t10` step).  Here a token
// lane walks a vertical STRIP of R rows at one (frame, column), as the forward kernel does: the window slides down, 3 new x loads
// + 1 dy load per token (R = 4: 5.5 loads per token with the two rows that prime the window, R = 8: 4.75).  Workgroup, partial
// layout, the reduction over the 32 token lanes and the finish kernel are v2's.
template <int R>
__global__ void __launch_bounds__(512)
dwconv3x3_wgrad3_kernel(const u16* __restrict__ x, const u16* __restrict__ dy, int B, int H, int W, int C8, int slices,
                        int strips_per_slice, float* __restrict__ partial) {
  __shared__ float red[8][32][8];  // 512 threads: 32 channel groups x 16 token lanes (80 accumulators + a 3 x 3 window of
                                   // 16-byte vectors per thread: 1024 threads would cap a thread at 128 registers and spill)
  const int c8l = threadIdx.x & 31, tl = threadIdx.x >> 5, wave = threadIdx.x >> 6;
  const int c8 = blockIdx.x * 32 + c8l, s = blockIdx.y;
  const int HS = (H + R - 1) / R;
  const long long strips = (long long)B * HS * W;
  const long long q0 = (long long)s * strips_per_slice, q1 = min(strips, q0 + strips_per_slice);
  float acc[10][8];
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[t][k] = 0.f;
  const long long rs = (long long)W * C8 * 8;  // elements per image row
  const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
  for (long long q = q0 + tl; q < q1; q += 16) {
    // strip q = (b, hs, w), w fastest: the 16 token lanes of a workgroup read neighbouring columns
    const int w = (int)(q % W);
    const long long t2 = q / W;
    const int hs = (int)(t2 % HS), b = (int)(t2 / HS);
    const int h0 = hs * R;
    const bool lok = w > 0, rok = w < W - 1;
    const u16* xc = x + (((long long)b * H * W + w) * C8 + c8) * 8;   // token (b, 0, w)
    const u16* gc = dy + (((long long)b * H * W + w) * C8 + c8) * 8;
    uint4 win[3][3];
    auto load_row = [&](int hh, uint4 (&row)[3]) {
      if (hh < 0 || hh >= H) { row[0] = row[1] = row[2] = zero; return; }
      const u16* p = xc + hh * rs;
      row[0] = lok ? *reinterpret_cast<const uint4*>(p - C8 * 8) : zero;
      row[1] = *reinterpret_cast<const uint4*>(p);
      row[2] = rok ? *reinterpret_cast<const uint4*>(p + C8 * 8) : zero;
    };
    load_row(h0 - 1, win[0]);
    load_row(h0, win[1]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int h = h0 + r;
      if (h >= H) break;
      load_row(h + 1, win[2]);
      float g[8];
      unpack8(*reinterpret_cast<const uint4*>(gc + h * rs), g);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[9][k] += g[k];
#pragma unroll
      for (int iy = 0; iy < 3; ++iy)
#pragma unroll
        for (int ix = 0; ix < 3; ++ix) {
          float xv[8];
          unpack8(win[iy][ix], xv);  // (a tap outside the map is a zero vector: it adds nothing, as v2's `continue`)
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[iy * 3 + ix][k] += g[k] * xv[k];
        }
#pragma unroll
      for (int ix = 0; ix < 3; ++ix) { win[0][ix] = win[1][ix]; win[1][ix] = win[2][ix]; }
    }
  }
  const int C = C8 * 8;
  float* o = partial + (long long)s * 10 * C;
#pragma unroll
  for (int t = 0; t < 10; ++t) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[t][k] += __shfl_xor(acc[t][k], 32, 64);  // the wave's two token lanes
    if ((threadIdx.x & 32) == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) red[wave][c8l][k] = acc[t][k];
    }
    __syncthreads();
    if (threadIdx.x < 256) {  // (channel group, k): fixed order over the 16 waves
      const int cl = threadIdx.x >> 3, k = threadIdx.x & 7;
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) v += red[q][cl][k];
      o[t * C + (blockIdx.x * 32 + cl) * 8 + k] = v;
    }
    __syncthreads();
  }
}

// dw[c][tap] = sum_s partial[s][tap][c], db[c] = sum_s partial[s][9][c]: the weight gradient in the parameter's layout [C][9]
// and the bias gradient in ONE launch (v1: two levels of the generic split-K reduce + a transpose copy).  Block = 32 channels
// x 8 slice lanes, grid = (C / 32, 10 taps); fixed summation order.
__global__ void __launch_bounds__(256)
dwconv3x3_wgrad_finish_kernel(const float* __restrict__ partial, int slices, int C, float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, t = blockIdx.y;
  float v = 0.f;
  if (c < C)
    for (int s = sl; s < slices; s += 8) v += partial[((long long)s * 10 + t) * C + c];
  red[sl][cl] = v;
  __syncthreads();
  if (sl == 0 && c < C) {
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) a += red[q][cl];
    if (t < 9) dw[(long long)c * 9 + t] = a;
    else if (db) db[c] = a;
  }
}

}  // namespace

extern "C" {

int combo_dwconv3x3_bf16(const void* x, const float* w_tap_major, const float* bias, int B, int H, int W, int C, int flip,
                         void* y, combo_stream_t stream) {
  if (!x || !w_tap_major || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0 || ((uintptr_t)x & 15) ||
      ((uintptr_t)y & 15) || ((uintptr_t)w_tap_major & 15) || ((uintptr_t)bias & 15))
    return COMBO_EINVAL;
  if ((long long)B * H * W * (C / 8) < (1ll << 31)) {  // strip kernel: 2.3 TB/s against 1.2 for the per-token kernel below
    if (H >= 32) {
      const long long total = (long long)B * ((H + 7) / 8) * W * (C / 8);
      hipLaunchKernelGGL(dwconv3x3_strip_kernel<8>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                         (const u16*)x, w_tap_major, bias, B, H, W, C / 8, flip, (u16*)y);
    } else {
      const long long total = (long long)B * ((H + 3) / 4) * W * (C / 8);
      hipLaunchKernelGGL(dwconv3x3_strip_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                         (const u16*)x, w_tap_major, bias, B, H, W, C / 8, flip, (u16*)y);
    }
    return (int)hipGetLastError();
  }
  const long long total = (long long)B * H * W * (C / 8);
  hipLaunchKernelGGL(dwconv3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const u16*)x, w_tap_major, bias, B, H, W, C / 8, flip, (u16*)y);
  return (int)hipGetLastError();
}

int combo_dwconv3x3_wgrad_finish_f32(const float* partials, int slices, int C, float* dw, float* db, combo_stream_t stream) {
  if (!partials || !dw || slices <= 0 || C <= 0) return COMBO_EINVAL;
  hipLaunchKernelGGL(dwconv3x3_wgrad_finish_kernel, dim3((C + 31) / 32, 10), dim3(256), 0, (hipStream_t)stream, partials, slices, C,
                     dw, db);
  return (int)hipGetLastError();
}

static bool wgrad_v2() { return true; }
static int g_wgrad_strips = 1;  // (v1 below stays for channel counts that are not a multiple of 256)

int combo_dwconv3x3_wgrad_strips(int on) {  // host-side switch of the launches that follow; returns the previous value
  const int prev = g_wgrad_strips;
  g_wgrad_strips = on ? 1 : 0;
  return prev;
}

int combo_dwconv3x3_wgrad_slices(int B, int H, int W, int C) {
  const long long tokens = (long long)B * H * W;
  if (wgrad_v2() && C % 256 == 0 && g_wgrad_strips) {  // v3: a token lane takes whole strips; a slice = a multiple of 16 strips (16 token lanes)
    const int R = H >= 32 ? 8 : 4;
    const long long strips = (long long)B * ((H + R - 1) / R) * W;
    long long s = 512 / (C / 256);
    const long long maxs = (strips + 15) / 16;
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    long long sps = (strips + s - 1) / s;
    sps = (sps + 15) / 16 * 16;
    s = (strips + sps - 1) / sps;
    return (int)(s < 1 ? 1 : s);
  }
  if (wgrad_v2() && C % 256 == 0) {  // v2: 32 channel groups x 32 token lanes per workgroup, ~512 workgroups
    long long s = 512 / (C / 256);
    if (s > tokens / 128) s = tokens / 128;  // at least 4 tokens per token lane
    return (int)(s < 1 ? 1 : s);
  }
  long long s = (65536LL * 8) / (C > 0 ? C : 8);  // ~64k threads
  if (s > tokens / 8) s = tokens / 8;             // at least 8 tokens per thread
  if (s < 1) s = 1;
  return (int)s;
}

int combo_dwconv3x3_wgrad_bf16(const void* x, const void* dy, int B, int H, int W, int C, int slices, float* partials,
                               combo_stream_t stream) {
  if (!x || !dy || !partials || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0 || slices <= 0 || ((uintptr_t)x & 15) ||
      ((uintptr_t)dy & 15) || ((uintptr_t)partials & 15))
    return COMBO_EINVAL;
  const long long tokens = (long long)B * H * W;
  const int tps = (int)((tokens + slices - 1) / slices);
  if (wgrad_v2() && C % 256 == 0) {
    if (g_wgrad_strips) {  // v3: strips of R rows per token lane (combo_dwconv3x3_wgrad_strips(0) brings v2 back, for A/B)
      const int R = H >= 32 ? 8 : 4;
      const long long strips = (long long)B * ((H + R - 1) / R) * W;
      int sps = (int)((strips + slices - 1) / slices);
      sps = (sps + 15) / 16 * 16;  // (as combo_dwconv3x3_wgrad_slices plans it; a caller's own slice count still covers every strip)
      if ((long long)sps * slices < strips) return COMBO_EINVAL;
      if (R == 8)
        hipLaunchKernelGGL(dwconv3x3_wgrad3_kernel<8>, dim3(C / 256, slices), dim3(512), 0, (hipStream_t)stream, (const u16*)x,
                           (const u16*)dy, B, H, W, C / 8, slices, sps, partials);
      else
        hipLaunchKernelGGL(dwconv3x3_wgrad3_kernel<4>, dim3(C / 256, slices), dim3(512), 0, (hipStream_t)stream, (const u16*)x,
                           (const u16*)dy, B, H, W, C / 8, slices, sps, partials);
      return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(dwconv3x3_wgrad2_kernel, dim3(C / 256, slices), dim3(1024), 0, (hipStream_t)stream, (const u16*)x,
                       (const u16*)dy, B, H, W, C / 8, slices, tps, partials);
    return (int)hipGetLastError();
  }
  const long long threads = (long long)slices * (C / 8);
  hipLaunchKernelGGL(dwconv3x3_wgrad_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const u16*)x, (const u16*)dy, B, H, W, C / 8, slices, tps, partials);
  return (int)hipGetLastError();
}

}  // extern "C"
