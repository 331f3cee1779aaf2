// MSDeformAttn backward, fused and windowed (D == 32, fp32): grad_value, grad_sampling_loc and grad_attn_weight from ONE
// launch in which value, grad_out, sampling_loc and attn_weight are each needed once.
//
// Replaces ms_deformable_col2im_cuda / ...col2im_gpu_kernel_shm_blocksize_aware_reduce_v1<32>
// (models/modeling/pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh:961-1331, :306-408; taps :92-164): one thread per
// (b, q, m, c), 4 global float atomics per tap and channel into grad_value, a serial thread-0 reduction for d/dloc, d/dw.
//
// Work decomposition: a workgroup owns (frame b, head m, WINDOW) where a window is a band of image rows [y0, y1) of ONE
// pyramid level - a whole level when its rows fit the LDS, otherwise equal bands (224 x 224: 7x7 | 14x14 | 28x28 rows 0-13 |
// rows 14-27; 512 x 512: 16x16 | 3 bands of 32x32 | 11 bands of 64x64).  A window keeps, for ALL 32 channels of the head:
//   * the value rows of its band plus one image row below it (the bottom taps of the band's last row)    - LDS, 128 B / row
//   * a fixed-point accumulator of grad_value for the rows of its band                                   - LDS, 128 B / row
// and scans the sampling points of its level of every query (P of the L*P points):
//   scatter  every tap that lands in the band adds  w_tap * a * grad_out[q, :]  to the accumulator row (LDS integer atomics:
//            bitwise deterministic, no float atomics anywhere - the reference's atomicAdd is neither);
//   gather   the window that OWNS a sample (the band holding its top tap row, clamped into the image) has the four tap rows in
//            LDS, computes the four <value_tap, grad_out[q]> products over all 32 channels and with them d/dw and d/dloc of
//            that sample - complete sums, written once with plain stores (the two-kernel design split the channels in halves,
//            read value / grad_out / loc / w twice and still needed the slab of the whole pyramid).
// So every sample is scattered by the window(s) its taps touch (no duplicated atomics) and differentiated by exactly one.
//
// Fixed point: two channels share one 64-bit LDS word, X = (v1 << 32) + sext(v0); sum(X) = 2^32 sum(v1) + sum(v0), hence
// low word = sum(v0) and high word = sum(v1) - [sum(v0) < 0], both exact as long as |sum| < 2^31.  A wave instruction
// ds_add_u64 covers 4 taps x 16 channel pairs = 4 rows x 128 contiguous bytes: conflict-free (6.4 cycles, tools/ubench;
// the 16-rows-x-4-lanes pattern of the previous kernel measured 9.5 cycles for a QUARTER of the channels).  Scales: per
// channel 1 / max_q |grad_out[q, c]|, per row 2^30 / W_r with W_r an upper bound of the row's total tap weight from a first,
// cheap pass over the same samples (geometry + one 4-byte LDS atomic per tap, no channel work).  A single window-wide scale
// 2^30 / sum |a| was measured: 2.3e-5 absolute error on grad_value at Lq = 1029 (5x that at 5376) - rows of the 7x7 level
// collect 300+ adds at a resolution set by a bound 50x above their real weight; with the row scales it is 6e-7.
//
// A wave handles 64 samples per iteration in three phases over a per-wave LDS record array:
//   A  lane = sample: tap geometry once -> record {(w_tap * scale, accumulator row) x 4 | lh, lw, a W, a H | slab rows}
//   B  lane = (sample of 8, 4 channels of 32): the owner's gather, 8 samples per step, DPP reductions over the 8 lanes
//   C  lane = (tap of 4, channel pair of 16): one ds_add_u64 per sample, skipped when no tap of the sample is in the band.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "combo_common.h"

namespace {

constexpr int kD = 32;
constexpr int kMaxLv = 8;
constexpr int kMaxWin = 48;
constexpr int kLds = 160 * 1024;

struct WinArgs {
  int n_win;
  int H[kMaxLv], W[kMaxLv], start[kMaxLv];
  short lvl[kMaxWin], y0[kMaxWin], y1[kMaxWin];
};

template <int CTRL>
__device__ __forceinline__ float dppf(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}

// float -> int, round to nearest (ties up): ONE instruction (v_cvt_rpi_i32_f32); __float2int_rn is v_rndne + v_cvt
__device__ __forceinline__ int cvt_rpi(float x) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

__device__ __forceinline__ void lds_dma16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// per-sample record arrays of one wave (64 samples): c0 {ws_k, accrow_k} x 4 (32 B), c1 {lh, lw, a*W, a*H} (16 B),
// c2 {slab rows 0|1, 2|3 as u16 pairs} (8 B)
constexpr int kRecBytes = 64 * (32 + 16 + 8);

template <int NW, int P>
__global__ void __launch_bounds__(NW * 64)
msda_bwd_win_d32(const float* __restrict__ gout, const float* __restrict__ value, const float* __restrict__ loc,
                 const float* __restrict__ aw, int B, int S, int M, int L, int Lq, float* __restrict__ gvalue,
                 float* __restrict__ gloc, float* __restrict__ gaw, WinArgs wa, unsigned long long* __restrict__ ts, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  combo_ts_begin(ts);
  constexpr int NT = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int logical = xcd_contiguous(blockIdx.x, gridDim.x);
  const int win = logical % wa.n_win;
  const int bm = logical / wa.n_win;
  const int m = bm % M, b = bm / M;
  const int lv = wa.lvl[win], y0 = wa.y0[win], y1 = wa.y1[win];
  const int H = wa.H[lv], W = wa.W[lv];
  const int R = (y1 - y0) * W;                 // accumulator rows of the band (local row R = sink of foreign taps)
  const int NR = R + (y1 < H ? W : 0);         // value rows in LDS: the band + one image row below (local row NR = zeros)
  const int row0 = wa.start[lv] + y0 * W;      // first pyramid row of the band
  const int LP = L * P;

  float* slab = reinterpret_cast<float*>(smem);                                       // [NR + 1][32] f32
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(slab + (NR + 1) * kD);  // [R + 1][16] u64
  int* wsum = reinterpret_cast<int*>(acc + (R + 1) * 16);                             // [R + 1] -> row scales
  float* red = reinterpret_cast<float*>(wsum + ((R + 1 + 3) & ~3));                   // [NW][36] + chmx[32] + misc[4]
  float* chmx = red + NW * 36;
  char* rec = reinterpret_cast<char*>(chmx + 36) + wave * kRecBytes;
  float2* c0 = reinterpret_cast<float2*>(rec);                 // [64][4] {ws, accrow}
  float4* c1 = reinterpret_cast<float4*>(rec + 64 * 32);       // [64]
  uint2* c2 = reinterpret_cast<uint2*>(rec + 64 * 48);         // [64]

  // ---- stage the value rows of the band (+ halo) with LDS-DMA; clear the accumulators --------------------------------------
  {
    const float* vb = value + (((long long)b * S + row0) * M + m) * kD + (lane & 7) * 4;
    for (int r0 = wave * 8; r0 < NR; r0 += NW * 8) {
      const int r = r0 + (lane >> 3);
      if (r < NR) lds_dma16(vb + (long long)r * M * kD, slab + r0 * kD);
    }
    if (tid < 8) *reinterpret_cast<float4*>(slab + NR * kD + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int i = tid; i < (R + 1) * 8; i += NT) reinterpret_cast<uint4*>(acc)[i] = make_uint4(0u, 0u, 0u, 0u);
  for (int r = tid; r <= R; r += NT) wsum[r] = 0;

  // ---- pre-pass: max_q |grad_out[q, c]| per channel, sum |a| over the level's samples ----------------------------------------
  {
    const int cg = tid & 7;
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
    bool nan = false;
    for (int q0 = tid >> 3; q0 < Lq; q0 += 4 * (NT / 8)) {  // 4 independent 16-byte loads in flight per lane
      float4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + u * (NT / 8);
        t[u] = q < Lq ? *reinterpret_cast<const float4*>(gout + (((long long)b * Lq + q) * M + m) * kD + cg * 4)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        mx[0] = fmaxf(mx[0], fabsf(t[u].x)); mx[1] = fmaxf(mx[1], fabsf(t[u].y));
        mx[2] = fmaxf(mx[2], fabsf(t[u].z)); mx[3] = fmaxf(mx[3], fabsf(t[u].w));
        nan |= !(t[u].x == t[u].x) || !(t[u].y == t[u].y) || !(t[u].z == t[u].z) || !(t[u].w == t[u].w);  // fmaxf drops NaNs
      }
    }
    float sa = 0.f;
    for (int i0 = tid; i0 < Lq * P; i0 += 4 * NT) {
      float t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * NT;
        const int q = i / P, p = i - q * P;
        t[u] = i < Lq * P ? aw[(((long long)b * Lq + q) * M + m) * LP + lv * P + p] : 0.f;
      }
      sa += (fabsf(t[0]) + fabsf(t[1])) + (fabsf(t[2]) + fabsf(t[3]));
    }
#pragma unroll
    for (int s = 8; s < 64; s <<= 1)
#pragma unroll
      for (int c = 0; c < 4; ++c) mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], s));
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) sa += __shfl_xor(sa, s);
    const bool wave_nan = __any(nan);
    if (lane < 8) {
#pragma unroll
      for (int c = 0; c < 4; ++c) red[wave * 36 + lane * 4 + c] = mx[c];
    }
    if (lane == 0) { red[wave * 36 + 32] = sa; red[wave * 36 + 33] = wave_nan ? 1.f : 0.f; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-DMA of the slab
  __syncthreads();
  if (tid < 32) {
    float mm = 0.f, tot = 0.f, bad = 0.f;
    for (int w = 0; w < NW; ++w) { mm = fmaxf(mm, red[w * 36 + tid]); tot += red[w * 36 + 32]; bad += red[w * 36 + 33]; }
    if (bad != 0.f || !(mm < 3.0e38f) || !(tot < 3.0e38f)) mm = __builtin_nanf("");  // non-finite grad_out -> NaN gradients
    chmx[tid] = mm;
    if (tid == 0) chmx[32] = tot;
  }
  __syncthreads();
  const float tot = chmx[32];
  const float wscale = tot > 0.f ? 1073741824.f / tot : 0.f;  // pass 0 accumulates |tap weight| at 2^30 / sum |a|
  const float inv_wscale = tot > 0.f ? tot * (1.f / 1073741824.f) : 0.f;

  const int n_samples = Lq * P;
  const int n_iter = (n_samples + 63) / 64;

  // tap geometry of sample i (reference .cuh:92-164): local rows of the four taps (slab / accumulator), weights
  auto fetch = [&](int i, float2& xy, float& a) {  // sampling location + attention weight of sample i of this level
    xy = make_float2(-4.f, -4.f);  // (outside every map)
    a = 0.f;
    if (i < n_samples) {
      const int q = i / P, p = i - q * P;
      const long long e = (((long long)b * Lq + q) * M + m) * LP + lv * P + p;
      xy = *reinterpret_cast<const float2*>(loc + e * 2);
      a = aw[e];
    }
  };
  auto geometry = [&](float2 xy, float a, bool live, int (&srow)[4], int (&arow)[4], float (&wt)[4], float4& par, bool& owner) {
    owner = false;
    par = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) { srow[k] = NR; arow[k] = R; wt[k] = 0.f; }
    if (!live) return;
    const float h_im = xy.y * H - 0.5f, w_im = xy.x * W - 0.5f;
    if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) {
      owner = y0 == 0;  // a sample outside the map has zero gradients: the level's first band writes them
      return;
    }
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int h0 = (int)hf, w0 = (int)wf;
    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
    const bool t_ok = h0 >= 0, b_ok = h0 + 1 <= H - 1, l_ok = w0 >= 0, r_ok = w0 + 1 <= W - 1;
    const int yo = h0 < 0 ? 0 : h0;
    owner = yo >= y0 && yo < y1;
    const int base = (h0 - y0) * W + w0;  // local row of the top-left tap
    const bool ok[4] = {t_ok && l_ok, t_ok && r_ok, b_ok && l_ok, b_ok && r_ok};
    const int rr[4] = {base, base + 1, base + W, base + W + 1};
    const float w4[4] = {hh * hw * a, hh * lw * a, lh * hw * a, lh * lw * a};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = h0 + (k >> 1);
      if (ok[k] && y >= y0 && y < y1) { arow[k] = rr[k]; wt[k] = w4[k]; }
      if (ok[k] && owner) srow[k] = rr[k];  // y in [y0, y1]: the band or its halo row
    }
    par = make_float4(lh, lw, a * W, a * H);
  };

  // ---- pass 0: W_r, an upper bound of the total tap weight a row can receive (two samples in flight per lane) ----------------------
  for (int it = wave; it < n_iter; it += 2 * NW) {
    float2 xy[2];
    float a[2];
    fetch(it * 64 + lane, xy[0], a[0]);
    fetch((it + NW) * 64 + lane, xy[1], a[1]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int srow[4], arow[4];
      float wt[4];
      float4 par;
      bool owner;
      geometry(xy[u], a[u], (it + u * NW) * 64 + lane < n_samples, srow, arow, wt, par, owner);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (arow[k] < R)
          __hip_atomic_fetch_add(wsum + arow[k], __float2int_ru(fabsf(wt[k]) * wscale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  float* rowscale = reinterpret_cast<float*>(wsum);
  for (int r = tid; r <= R; r += NT) {
    const float wr = (float)wsum[r] * inv_wscale * 1.0001f;
    rowscale[r] = (r < R && wr > 0.f) ? 1073741824.f / wr : 0.f;
  }
  __syncthreads();

  // ---- the pass over the samples --------------------------------------------------------------------------------------------------------
  const int kk = lane >> 4, cp = lane & 15;    // phase C: tap, channel pair
  const int g8 = lane >> 3, cg = lane & 7;     // phase B: sample of 8, 4 channels
  const float inv0 = [&] { const float mm = chmx[2 * cp]; return mm > 0.f ? 1.f / mm : (mm == mm ? 0.f : mm); }();
  const float inv1 = [&] { const float mm = chmx[2 * cp + 1]; return mm > 0.f ? 1.f / mm : (mm == mm ? 0.f : mm); }();
  constexpr int QI = 64 / P;  // queries per wave iteration (64 % P == 0: an iteration starts on a query boundary)
  const bool full_level = y0 == 0 && y1 == H;  // every sample inside the map touches the window: no per-sample skip test
  float2 xy_next;
  float a_next;
  fetch(wave * 64 + lane, xy_next, a_next);
  for (int it = wave; it < n_iter; it += NW) {
    const int ibase = it * 64;
    const float2 xy_cur = xy_next;
    const float a_cur = a_next;
    fetch((it + NW) * 64 + lane, xy_next, a_next);  // the next iteration's sample: in flight during this one
    const int qbase = ibase / P;
    // grad_out rows of the iteration's queries, requested first so that their latency hides behind phase A:
    //   phase C: this lane's channel pair of every query;  phase B: this lane's 4 channels of the queries of its 8 samples
    float2 gq[QI];
    float4 tgB[8];
    auto prefetch = [&]() {
#pragma unroll
      for (int u = 0; u < QI; ++u) {
        const int q = qbase + u < Lq ? qbase + u : Lq - 1;
        const float2 t = *reinterpret_cast<const float2*>(gout + (((long long)b * Lq + q) * M + m) * kD + 2 * cp);
        gq[u] = make_float2(t.x * inv0, t.y * inv1);  // normalised to [-1, 1] per channel
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int qq = qbase + (j * 8 + g8) / P;
        const int q = qq < Lq ? qq : Lq - 1;
        tgB[j] = *reinterpret_cast<const float4*>(gout + (((long long)b * Lq + q) * M + m) * kD + cg * 4);
      }
    };
    // a whole-level window works on every iteration: request the rows first, their latency hides behind phase A; a band sees
    // most iterations pass by untouched (queries are in raster order, offsets are local) and requests them once it knows
    if (full_level) prefetch();
    // -- phase A: one lane per sample
    unsigned long long own_mask, touch_mask;
    {
      const int i = ibase + lane;
      int srow[4], arow[4];
      float wt[4];
      float4 par;
      bool owner;
      geometry(xy_cur, a_cur, i < n_samples, srow, arow, wt, par, owner);
      own_mask = __ballot(owner);
      touch_mask = __ballot(arow[0] < R || arow[1] < R || arow[2] < R || arow[3] < R);
      if (dbg & 1) own_mask = 0;    // ablation (COMBO_MSDA_BWD_DBG): no gather phase
      if (dbg & 2) touch_mask = 0;  // ablation: no scatter phase
      if (!full_level && (own_mask | touch_mask) == 0ull) continue;  // nothing of this band in these 64 samples (wave-uniform)
      float4 lo, hi;
      lo.x = wt[0] * rowscale[arow[0]]; lo.y = __int_as_float(arow[0]);
      lo.z = wt[1] * rowscale[arow[1]]; lo.w = __int_as_float(arow[1]);
      hi.x = wt[2] * rowscale[arow[2]]; hi.y = __int_as_float(arow[2]);
      hi.z = wt[3] * rowscale[arow[3]]; hi.w = __int_as_float(arow[3]);
      reinterpret_cast<float4*>(c0)[lane * 2] = lo;
      reinterpret_cast<float4*>(c0)[lane * 2 + 1] = hi;
      c1[lane] = par;
      c2[lane] = make_uint2((unsigned)srow[0] | ((unsigned)srow[1] << 16), (unsigned)srow[2] | ((unsigned)srow[3] << 16));
    }
    if (!full_level) prefetch();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // -- phase B: the owner's gather -> d out / d w, d out / d loc (8 samples per step, 8 lanes x 4 channels per sample)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (((own_mask >> (8 * j)) & 0xffull) == 0ull) continue;  // wave-uniform
      const int s = j * 8 + g8;
      const int i = ibase + s;
      const int ic = i < n_samples ? i : n_samples - 1;
      const int q = ic / P, p = ic - q * P;
      const long long qm = ((long long)b * Lq + q) * M + m;
      const float4 tg = tgB[j];
      const float4 pp = c1[s];
      const uint2 o = c2[s];
      const float* sl = slab + cg * 4;
      const float4 v0 = *reinterpret_cast<const float4*>(sl + (o.x & 0xffffu) * kD);
      const float4 v1 = *reinterpret_cast<const float4*>(sl + (o.x >> 16) * kD);
      const float4 v2 = *reinterpret_cast<const float4*>(sl + (o.y & 0xffffu) * kD);
      const float4 v3 = *reinterpret_cast<const float4*>(sl + (o.y >> 16) * kD);
      const float d0 = tg.x * v0.x + tg.y * v0.y + tg.z * v0.z + tg.w * v0.w;
      const float d1 = tg.x * v1.x + tg.y * v1.y + tg.z * v1.z + tg.w * v1.w;
      const float d2 = tg.x * v2.x + tg.y * v2.y + tg.z * v2.z + tg.w * v2.w;
      const float d3 = tg.x * v3.x + tg.y * v3.y + tg.z * v3.z + tg.w * v3.w;
      const float lh = pp.x, lw = pp.y, hh = 1.f - lh, hw = 1.f - lw;
      float sw = hh * hw * d0 + hh * lw * d1 + lh * hw * d2 + lh * lw * d3;  // d out / d w
      float sy = (-hw * d0 - lw * d1 + hw * d2 + lw * d3) * pp.w;              // * a * H  (.cuh:162-163)
      float sx = (-hh * d0 + hh * d1 - lh * d2 + lh * d3) * pp.z;              // * a * W
      sw += dppf<0xB1>(sw); sx += dppf<0xB1>(sx); sy += dppf<0xB1>(sy);     // quad_perm [1,0,3,2]
      sw += dppf<0x4E>(sw); sx += dppf<0x4E>(sx); sy += dppf<0x4E>(sy);     // quad_perm [2,3,0,1]
      sw += dppf<0x141>(sw); sx += dppf<0x141>(sx); sy += dppf<0x141>(sy);  // row_half_mirror
      if (cg == 0 && ((own_mask >> s) & 1ull)) {
        const long long e = qm * LP + lv * P + p;
        gaw[e] = sw;
        *reinterpret_cast<float2*>(gloc + e * 2) = make_float2(sx, sy);
      }
    }
    // -- phase C: scatter, one ds_add_u64 wave instruction per sample (4 taps x 16 channel pairs); the records of 8 samples are
    //    read ahead of their 8 atomics (the wave-uniform skips would otherwise serialise read -> wait -> add per sample)
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (((touch_mask >> (8 * g)) & 0xffull) == 0ull) continue;  // wave-uniform
      float2 wr[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wr[u] = c0[(g * 8 + u) * 4 + kk];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = g * 8 + u;
        if (!full_level && !((touch_mask >> s) & 1ull)) continue;  // wave-uniform (a skipped sample of a full level adds 0 to the sink row)
        const float g0 = gq[s / P].x, g1 = gq[s / P].y;
        const int ar = __float_as_int(wr[u].y);
        const int v0 = cvt_rpi(wr[u].x * g0), v1 = cvt_rpi(wr[u].x * g1);
        const unsigned long long x = ((unsigned long long)(unsigned)(v1 + (v0 >> 31)) << 32) | (unsigned)v0;
        __hip_atomic_fetch_add(acc + ar * 16 + cp, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  // ---- flush: grad_value rows of the band, all 32 channels, plain coalesced stores -------------------------------------------------
  {
    const int c4 = tid & 7;  // channels 4 c4 .. 4 c4 + 3 = pairs 2 c4, 2 c4 + 1
    const float m0 = chmx[4 * c4], m1 = chmx[4 * c4 + 1], m2 = chmx[4 * c4 + 2], m3 = chmx[4 * c4 + 3];
    float* gb = gvalue + (((long long)b * S + row0) * M + m) * kD + c4 * 4;
    for (int r = tid >> 3; r < R; r += NT / 8) {
      const ulonglong2 w2 = *reinterpret_cast<const ulonglong2*>(acc + r * 16 + 2 * c4);
      const int a0 = (int)(unsigned)w2.x, a1 = (int)(unsigned)(w2.x >> 32) + (a0 < 0 ? 1 : 0);
      const int a2 = (int)(unsigned)w2.y, a3 = (int)(unsigned)(w2.y >> 32) + (a2 < 0 ? 1 : 0);
      const float rs = rowscale[r];
      const float f = rs > 0.f ? 1.f / rs : 0.f;
      *reinterpret_cast<float4*>(gb + (long long)r * M * kD) =
          make_float4((float)a0 * f * m0, (float)a1 * f * m1, (float)a2 * f * m2, (float)a3 * f * m3);
    }
  }
  combo_ts_end(ts);
}

inline size_t win_lds_bytes(int R, int halo, int nw) {
  return (size_t)(R + halo + 1) * 128 + (size_t)(R + 1) * 128 + (size_t)((R + 1 + 3) & ~3) * 4 + (size_t)(nw * 36 + 36) * 4 +
         (size_t)nw * kRecBytes + 16;
}

}  // namespace

extern "C" {

// 1 when combo_msda_backward_win_f32 takes this geometry (every output element is then written: no zero-fill needed)
int combo_msda_backward_win_ok(const int* host_shapes, int L, int P, int D, int elem_bytes) {
  if (!host_shapes || elem_bytes != 4 || D != kD || L <= 0 || L > kMaxLv || P != 4) return 0;  // (P: template instance)
  int wins = 0;
  for (int l = 0; l < L; ++l) {
    const int H = host_shapes[2 * l], W = host_shapes[2 * l + 1];
    if (H <= 0 || W <= 0 || H > 32767 || W > 4096) return 0;
    if (win_lds_bytes(W, W, 8) > (size_t)kLds) return 0;  // not even one image row + halo fits
    long long rmax = 0;
    for (int rows = 1; rows <= H; ++rows)
      if (win_lds_bytes(rows * W, rows < H ? W : 0, 8) <= (size_t)kLds) rmax = rows; else break;
    if ((long long)rmax * W > 60000) return 0;  // 16-bit local rows
    wins += (int)((H + rmax - 1) / rmax);
  }
  return wins <= kMaxWin ? 1 : 0;
}

/* Fused, windowed MSDeformAttn backward (D == 32, fp32).  Same operands as combo_msda_backward_f32 plus the level geometry ON
 * THE HOST (host_shapes [L,2] ints = spatial_shapes, host_start [L] = level_start_index): the window table and the LDS budget
 * are functions of the level sizes, and the reference's launcher receives them as device tensors only
 * (ms_deform_attn_cuda.cu:72-73).  Writes every element of the three gradients. */
int combo_msda_backward_win_f32(const float* grad_out, const float* value, const int* host_shapes, const int* host_start,
                                const float* sampling_loc, const float* attn_weight, int B, int S, int M, int D, int L, int Lq,
                                int P, float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                                combo_stream_t stream) {
  if (!grad_out || !value || !host_shapes || !host_start || !sampling_loc || !attn_weight || !grad_value || !grad_sampling_loc ||
      !grad_attn_weight || B <= 0 || S <= 0 || M <= 0 || Lq <= 0)
    return COMBO_EINVAL;
  if (!combo_msda_backward_win_ok(host_shapes, L, P, D, 4)) return COMBO_EINVAL;
  static const int nw_env = [] { const char* e = getenv("COMBO_MSDA_BWD_WAVES"); return e ? atoi(e) : 0; }();
  static const int per_cu = [] { const char* e = getenv("COMBO_MSDA_BWD_PER_CU"); const int v = e ? atoi(e) : 1; return v >= 1 && v <= 4 ? v : 1; }();
  static const int cap_kb = [] { const char* e = getenv("COMBO_MSDA_BWD_LDS_KB"); return e ? atoi(e) : 0; }();
  const size_t lds_cap = cap_kb > 0 ? (size_t)cap_kb * 1024 : (size_t)kLds / per_cu;
  static const int dbg = [] { const char* e = getenv("COMBO_MSDA_BWD_DBG"); return e ? atoi(e) : 0; }();  // ablation bits
  WinArgs wa;
  wa.n_win = 0;
  size_t lds_max = 0;
  long long rows_total = 0;
  int nw = (nw_env == 4 || nw_env == 6 || nw_env == 8 || nw_env == 12) ? nw_env : 12;
  for (int pass = 0; pass < 2; ++pass) {
    // pass 0 with the preferred wave count; if a level's single image row does not fit next to 12 waves of records, 8 waves
    wa.n_win = 0;
    lds_max = 0;
    rows_total = 0;
    bool ok = true;
    for (int l = 0; l < L && ok; ++l) {
      const int H = host_shapes[2 * l], W = host_shapes[2 * l + 1];
      wa.H[l] = H; wa.W[l] = W; wa.start[l] = host_start[l];
      rows_total += (long long)H * W;
      int rmax = 0;
      for (int rows = 1; rows <= H; ++rows)
        if (win_lds_bytes(rows * W, rows < H ? W : 0, nw) <= lds_cap) rmax = rows; else break;
      if (rmax == 0) { ok = false; break; }
      const int bands = (H + rmax - 1) / rmax;
      const int per = (H + bands - 1) / bands;
      for (int y = 0; y < H; y += per) {
        if (wa.n_win >= kMaxWin) { ok = false; break; }
        const int ye = y + per < H ? y + per : H;
        wa.lvl[wa.n_win] = (short)l; wa.y0[wa.n_win] = (short)y; wa.y1[wa.n_win] = (short)ye;
        const size_t need = win_lds_bytes((ye - y) * W, ye < H ? W : 0, nw);
        lds_max = need > lds_max ? need : lds_max;
        ++wa.n_win;
      }
    }
    if (ok) break;
    if (pass == 1 || nw <= 8) return COMBO_EINVAL;
    nw = 8;
  }
  if (rows_total != S) return COMBO_EINVAL;
  const long long grid = (long long)B * M * wa.n_win;
  if (grid > 0x7fffffffLL) return COMBO_EINVAL;
  // algorithmic bytes: value, grad_out, loc, w read once; the three gradients written once
  const double bytes = 4.0 * B * (2.0 * ((double)S + Lq) * M * kD + 2.0 * 3.0 * (double)Lq * M * L * P);
  unsigned long long* ts = combo_timing_next_slot(COMBO_TS_MSDA_BWD, bytes, bytes);
  hipError_t e = hipSuccess;
#define COMBO_LAUNCH_WIN(NWV)                                                                                                     \
  do {                                                                                                                            \
    static bool attr = false;                                                                                                     \
    if (!attr) {                                                                                                                  \
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_bwd_win_d32<NWV, 4>),                                            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLds);                                                  \
      if (e != hipSuccess) return (int)e;                                                                                         \
      attr = true;                                                                                                                \
    }                                                                                                                             \
    hipLaunchKernelGGL((msda_bwd_win_d32<NWV, 4>), dim3((unsigned)grid), dim3(NWV * 64), lds_max, (hipStream_t)stream, grad_out,  \
                       value, sampling_loc, attn_weight, B, S, M, L, Lq, grad_value, grad_sampling_loc, grad_attn_weight, wa,     \
                       ts, dbg);                                                                                                  \
  } while (0)
  if (nw == 12) COMBO_LAUNCH_WIN(12);
  else if (nw == 8) COMBO_LAUNCH_WIN(8);
  else if (nw == 6) COMBO_LAUNCH_WIN(6);
  else COMBO_LAUNCH_WIN(4);
#undef COMBO_LAUNCH_WIN
  return (int)hipGetLastError();
}

}  // extern "C"
