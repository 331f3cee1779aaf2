// MSDeformAttn core op for MI355X (gfx950, wave64, 160 KiB LDS/CU).
//
// Replaces the reference launchers ms_deformable_im2col_cuda / ms_deformable_col2im_cuda
// (models/modeling/pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh:928-959, 961-1331).
// Semantics follow the reference device functions (.cuh:38-89 forward taps, :92-164 backward taps):
//   h_im = loc_y*H - 0.5, w_im = loc_x*W - 0.5, a sample contributes only if
//   h_im > -1 && w_im > -1 && h_im < H && w_im < W; the four taps are individually bounds-checked.
//
// Two families of kernels:
//   * generic  - any D/L/P/S, float or double.  One lane owns CH consecutive channels of one (b,q,m)
//                and gathers 16-byte pieces of the value rows straight from L2 (a row = D channels of
//                one head is contiguous).  Backward reduces d/dloc, d/dw over the D/CH lanes of a
//                (b,q,m) group with DPP shuffles and uses one hardware float atomic per tap/channel.
//   * LDS      - D == 32, fp32, the whole value slab of one (frame, head) — S rows x 128 B — is staged
//                once in the CU's LDS (S=1029 at 224x224 -> 131.7 KB) and all 48 taps/query are served
//                from LDS with ds_read_b128.  Forward splits the work of a wave in a coordinate phase
//                (one lane per (query, point): 4 row indices + 4 weights -> per-wave LDS scratch) and
//                a gather phase (8 lanes x 4 channels per query).  Backward stages one 16-channel half
//                of the slab plus a same-sized gradient accumulator in LDS, scatters with LDS float
//                atomics (no global atomics on grad_value) and flushes the accumulator with plain
//                coalesced stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "combo_common.h"

namespace {

constexpr int kMaxLevels = 8;

template <typename T> struct VecOf;
template <> struct VecOf<float> { using v4 = float4; };
template <> struct VecOf<double> { using v4 = double4; };

// -------------------------------------------------------------------------------------------------
// generic forward: one lane = CH channels of one (b, q, m)
// -------------------------------------------------------------------------------------------------
template <typename T, int CH>
__global__ void __launch_bounds__(256)
msda_fwd_generic(const T* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                 const T* __restrict__ loc, const T* __restrict__ aw, int B, int S, int M, int D, int L, int Lq, int P,
                 T* __restrict__ out, long long total) {
  const int DC = D / CH;
  const long long rs = (long long)M * D;  // row stride of `value` in elements
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % DC);
    long long t = idx / DC;  // (b*Lq + q)*M + m
    const int m = (int)(t % M);
    const long long bq = t / M;
    const int b = (int)(bq / Lq);
    const T* lp = loc + t * (long long)L * P * 2;
    const T* wp = aw + t * (long long)L * P;
    const T* vb = value + (long long)b * S * rs + (long long)m * D + cg * CH;
    T acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = 0;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const T* vl = vb + (long long)lsi[l] * rs;
      for (int p = 0; p < P; ++p) {
        const T x = lp[(l * P + p) * 2], y = lp[(l * P + p) * 2 + 1];
        const T a = wp[l * P + p];
        const T h_im = y * H - (T)0.5, w_im = x * W - (T)0.5;
        if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
          const int h0 = (int)floor(h_im), w0 = (int)floor(w_im);
          const T lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;
          const T wt[4] = {hh * hw * a, hh * lw * a, lh * hw * a, lh * lw * a};
          const bool ok[4] = {h0 >= 0 && w0 >= 0, h0 >= 0 && w0 + 1 <= W - 1, h0 + 1 <= H - 1 && w0 >= 0,
                              h0 + 1 <= H - 1 && w0 + 1 <= W - 1};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (ok[k]) {
              const T* src = vl + ((long long)(h0 + (k >> 1)) * W + (w0 + (k & 1))) * rs;
              if constexpr (CH == 4) {
                const typename VecOf<T>::v4 v = *reinterpret_cast<const typename VecOf<T>::v4*>(src);
                acc[0] += wt[k] * v.x; acc[1] += wt[k] * v.y; acc[2] += wt[k] * v.z; acc[3] += wt[k] * v.w;
              } else {
#pragma unroll
                for (int c = 0; c < CH; ++c) acc[c] += wt[k] * src[c];
              }
            }
          }
        }
      }
    }
    T* o = out + t * D + cg * CH;
    if constexpr (CH == 4) {
      typename VecOf<T>::v4 r; r.x = acc[0]; r.y = acc[1]; r.z = acc[2]; r.w = acc[3];
      *reinterpret_cast<typename VecOf<T>::v4*>(o) = r;
    } else {
#pragma unroll
      for (int c = 0; c < CH; ++c) o[c] = acc[c];
    }
  }
}

// -------------------------------------------------------------------------------------------------
// generic backward.  G = D/CH lanes form one (b,q,m) group (G a power of two <= 64, lanes of a group are
// adjacent in the wave so xor-shuffles reduce inside it).  CH == 4.
// -------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void atomic_add_relaxed(T* p, T v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T, int G, bool VALUE = true>
__global__ void __launch_bounds__(256)
msda_bwd_generic(const T* __restrict__ gout, const T* __restrict__ value, const int64_t* __restrict__ shapes,
                 const int64_t* __restrict__ lsi, const T* __restrict__ loc, const T* __restrict__ aw, int B, int S,
                 int M, int D, int L, int Lq, int P, T* __restrict__ gvalue, T* __restrict__ gloc, T* __restrict__ gaw,
                 long long total) {
  constexpr int CH = 4;
  const long long rs = (long long)M * D;
  // total is padded to a multiple of blockDim so that every lane of a group takes part in the shuffles
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % G);
    const long long t = idx / G;
    const bool live = t < (long long)B * Lq * M;
    const long long tt = live ? t : 0;
    const int m = (int)(tt % M);
    const int b = (int)((tt / M) / Lq);
    const T* lp = loc + tt * (long long)L * P * 2;
    const T* wp = aw + tt * (long long)L * P;
    const long long voff = (long long)b * S * rs + (long long)m * D + cg * CH;
    const typename VecOf<T>::v4 tg4 = *reinterpret_cast<const typename VecOf<T>::v4*>(gout + tt * D + cg * CH);
    const T tg[4] = {live ? tg4.x : 0, live ? tg4.y : 0, live ? tg4.z : 0, live ? tg4.w : 0};
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const long long lo = voff + (long long)lsi[l] * rs;
      for (int p = 0; p < P; ++p) {
        const T x = lp[(l * P + p) * 2], y = lp[(l * P + p) * 2 + 1];
        const T a = wp[l * P + p];
        const T h_im = y * H - (T)0.5, w_im = x * W - (T)0.5;
        T g_w = 0, g_x = 0, g_y = 0;
        if (live && h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
          const int h0 = (int)floor(h_im), w0 = (int)floor(w_im);
          const T lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;
          const T wt[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
          const T dh[4] = {-hw, -lw, hw, lw};  // d(weight_k)/d(h)   (.cuh:129-153)
          const T dw[4] = {-hh, hh, -lh, lh};  // d(weight_k)/d(w)
          const bool ok[4] = {h0 >= 0 && w0 >= 0, h0 >= 0 && w0 + 1 <= W - 1, h0 + 1 <= H - 1 && w0 >= 0,
                              h0 + 1 <= H - 1 && w0 + 1 <= W - 1};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (ok[k]) {
              const long long o = lo + ((long long)(h0 + (k >> 1)) * W + (w0 + (k & 1))) * rs;
              const typename VecOf<T>::v4 v4 = *reinterpret_cast<const typename VecOf<T>::v4*>(value + o);
              const T v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                g_w += tg[c] * wt[k] * v[c];
                g_y += tg[c] * dh[k] * v[c];
                g_x += tg[c] * dw[k] * v[c];
                if constexpr (VALUE) atomic_add_relaxed(gvalue + o + c, wt[k] * tg[c] * a);
              }
            }
          }
          g_x *= a * W;  // .cuh:162-163
          g_y *= a * H;
        }
#pragma unroll
        for (int s = 1; s < G; s <<= 1) {
          g_w += __shfl_xor(g_w, s);
          g_x += __shfl_xor(g_x, s);
          g_y += __shfl_xor(g_y, s);
        }
        if (live && cg == 0) {
          gaw[tt * (long long)L * P + l * P + p] = g_w;
          gloc[(tt * (long long)L * P + l * P + p) * 2] = g_x;
          gloc[(tt * (long long)L * P + l * P + p) * 2 + 1] = g_y;
        }
      }
    }
  }
}

// any D: one thread per (b,q,m,l,p); serial over channels.  Slow; used only for odd D (the reference's
// gradcheck cases D in {30, 71, 1025, ...}, ops/test.py:95-96).
template <typename T>
__global__ void __launch_bounds__(256)
msda_bwd_serial(const T* __restrict__ gout, const T* __restrict__ value, const int64_t* __restrict__ shapes,
                const int64_t* __restrict__ lsi, const T* __restrict__ loc, const T* __restrict__ aw, int B, int S,
                int M, int D, int L, int Lq, int P, T* __restrict__ gvalue, T* __restrict__ gloc, T* __restrict__ gaw,
                long long total) {
  const long long rs = (long long)M * D;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(idx % P);
    const int l = (int)((idx / P) % L);
    const long long t = idx / ((long long)P * L);
    const int m = (int)(t % M);
    const int b = (int)((t / M) / Lq);
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const T x = loc[idx * 2], y = loc[idx * 2 + 1], a = aw[idx];
    const T h_im = y * H - (T)0.5, w_im = x * W - (T)0.5;
    T g_w = 0, g_x = 0, g_y = 0;
    if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
      const int h0 = (int)floor(h_im), w0 = (int)floor(w_im);
      const T lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;
      const T wt[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
      const T dh[4] = {-hw, -lw, hw, lw};
      const T dw[4] = {-hh, hh, -lh, lh};
      const bool ok[4] = {h0 >= 0 && w0 >= 0, h0 >= 0 && w0 + 1 <= W - 1, h0 + 1 <= H - 1 && w0 >= 0,
                          h0 + 1 <= H - 1 && w0 + 1 <= W - 1};
      const long long lo = (long long)b * S * rs + (long long)m * D + (long long)lsi[l] * rs;
      const T* tg = gout + t * D;
      for (int k = 0; k < 4; ++k) {
        if (!ok[k]) continue;
        const long long o = lo + ((long long)(h0 + (k >> 1)) * W + (w0 + (k & 1))) * rs;
        for (int c = 0; c < D; ++c) {
          const T v = value[o + c];
          g_w += tg[c] * wt[k] * v;
          g_y += tg[c] * dh[k] * v;
          g_x += tg[c] * dw[k] * v;
          atomic_add_relaxed(gvalue + o + c, wt[k] * tg[c] * a);
        }
      }
      g_x *= a * W;
      g_y *= a * H;
    }
    gaw[idx] = g_w;
    gloc[idx * 2] = g_x;
    gloc[idx * 2 + 1] = g_y;
  }
}

// -------------------------------------------------------------------------------------------------
// LDS-staged forward, D == 32, fp32.
//   grid  = B*M*QT workgroups (QT query tiles per (frame, head)), block = NW waves (NW = 8 or 12,
//           whatever fits next to the slab)
//   LDS   = slab (S+1 rows x 128 B, row S is all-zero and is where out-of-range taps point)
//           + NW waves x QW*LP x (float4 weights + uint2 packed row indices)
//   staging uses the LDS-DMA path (global_load_lds_dwordx4: 8 rows = 1 KiB per wave instruction, no
//   VGPR round trip, all requests in flight at once).
// -------------------------------------------------------------------------------------------------
constexpr int kQW = 8;       // queries per wave iteration (8 lanes x 4 channels each in the gather phase)
constexpr int kMaxLP = 16;   // L*P supported by the LDS kernels
constexpr int kFwdPre = (kQW * kMaxLP + 63) / 64;

__device__ __forceinline__ int xcd_remap(int id, int n) {
  // blocks are dispatched round-robin over the 8 XCDs (private L2s): give each XCD a contiguous chunk
  // of the logical index space so that the heads / query tiles of one frame share an L2.  Bijective.
  const int q = n >> 3, r = n & 7, xcd = id & 7, j = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ void dma16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__global__ void __launch_bounds__(768)
msda_fwd_lds_d32(const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                 const float* __restrict__ loc, const float* __restrict__ aw, int B, int S, int M, int L, int Lq, int P,
                 int QT, float* __restrict__ out, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = 32;
  float* slab = reinterpret_cast<float*>(smem);
  const int LP = L * P;
  const int NW = blockDim.x >> 6;
  const int slab_bytes = (S + 1) * D * 4;
  float4* wts_all = reinterpret_cast<float4*>(smem + slab_bytes);
  uint2* offs_all = reinterpret_cast<uint2*>(smem + slab_bytes + NW * kQW * LP * 16);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = logical % QT;
  const int bm = logical / QT;
  const int m = bm % M, b = bm / M;

  // ---- stage the (b, m) slab: rows of 128 B at stride M*D*4 in global -> contiguous rows in LDS ----
  if (dbg != 2) {
    const float* vb = value + ((long long)b * S * M + m) * D + (lane & 7) * 4;
    for (int r0 = wave * 8; r0 < S; r0 += NW * 8) {  // 8 rows per wave instruction
      const int r = r0 + (lane >> 3);
      if (r < S) dma16(vb + (long long)r * M * D, slab + r0 * D);
    }
    if (tid < 8) *reinterpret_cast<float4*>(slab + S * D + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // level table (wave-uniform values; tiny) - loaded while the DMA is in flight
  int lvH[kMaxLevels], lvW[kMaxLevels], lvS[kMaxLevels];
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l) {
    lvH[l] = l < L ? (int)shapes[2 * l] : 1;
    lvW[l] = l < L ? (int)shapes[2 * l + 1] : 1;
    lvS[l] = l < L ? (int)lsi[l] : 0;
  }

  float4* wts = wts_all + wave * kQW * LP;
  uint2* offs = offs_all + wave * kQW * LP;
  const int qbeg = (int)(((long long)Lq * qt) / QT), qend = (int)(((long long)Lq * (qt + 1)) / QT);
  const int npairs = kQW * LP;

  // Per-lane constants of the coordinate phase.  A lane always serves the same (query slot, point) pairs
  // i = lane + 64*j, so the decomposition i -> (slot, point, level) and the level geometry are loop invariant:
  // computing them here keeps three runtime integer divisions (~40 VALU instructions each) out of the hot loop.
  int c_ql[kFwdPre], c_H[kFwdPre], c_W[kFwdPre], c_st[kFwdPre];
  long long c_eoff[kFwdPre];  // element offset of (slot, point) relative to the first query of the iteration
  bool c_on[kFwdPre];
#pragma unroll
  for (int j = 0; j < kFwdPre; ++j) {
    const int i = lane + j * 64;
    const int ql = i / LP, pt = i - ql * LP;
    const int l = pt / P;
    c_on[j] = i < npairs;
    c_ql[j] = ql;
    c_eoff[j] = (long long)ql * M * LP + pt;
    int H = lvH[0], W = lvW[0], st = lvS[0];
#pragma unroll
    for (int k = 1; k < kMaxLevels; ++k)
      if (l == k) { H = lvH[k]; W = lvW[k]; st = lvS[k]; }
    c_H[j] = H; c_W[j] = W; c_st[j] = st;
  }
  // software prefetch of the sampling locations / weights of the next wave iteration
  float2 pxy[kFwdPre];
  float pa[kFwdPre];
  auto prefetch = [&](int q0) {
    const long long base = (((long long)b * Lq + q0) * M + m) * LP;
#pragma unroll
    for (int j = 0; j < kFwdPre; ++j) {
      pxy[j] = make_float2(-8.f, -8.f);
      pa[j] = 0.f;
      if (c_on[j] && q0 + c_ql[j] < qend) {
        const long long e = base + c_eoff[j];
        pxy[j] = *reinterpret_cast<const float2*>(loc + e * 2);
        pa[j] = aw[e];
      }
    }
  };
  int q0 = qbeg + wave * kQW;
  if (q0 < qend) prefetch(q0);
  __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and publishes the slab
  if (dbg == 1) return;

  for (; q0 < qend; q0 += NW * kQW) {
    // ---- coordinate phase: one lane per (query, point) ----
#pragma unroll
    for (int j = 0; j < kFwdPre; ++j) {
      const int i = lane + j * 64;
      if (c_on[j] && dbg != 3) {
        float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned r01 = (unsigned)S | ((unsigned)S << 16), r23 = r01;
        const float2 xy = pxy[j];
        const float a = pa[j];
        const int H = c_H[j], W = c_W[j], st = c_st[j];
        const float h_im = xy.y * H - 0.5f, w_im = xy.x * W - 0.5f;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
          const float hf = floorf(h_im), wf = floorf(w_im);
          const int h0 = (int)hf, w0 = (int)wf;
          const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
          const bool t_ok = h0 >= 0, b_ok = h0 + 1 <= H - 1, l_ok = w0 >= 0, r_ok = w0 + 1 <= W - 1;
          const int base = st + h0 * W + w0;
          const unsigned r0 = (t_ok && l_ok) ? (unsigned)base : (unsigned)S;
          const unsigned r1 = (t_ok && r_ok) ? (unsigned)(base + 1) : (unsigned)S;
          const unsigned r2 = (b_ok && l_ok) ? (unsigned)(base + W) : (unsigned)S;
          const unsigned r3 = (b_ok && r_ok) ? (unsigned)(base + W + 1) : (unsigned)S;
          r01 = r0 | (r1 << 16);
          r23 = r2 | (r3 << 16);
          wv = make_float4(hh * hw * a, hh * lw * a, lh * hw * a, lh * lw * a);
        }
        wts[i] = wv;
        offs[i] = make_uint2(r01, r23);
      }
    }
    if (q0 + NW * kQW < qend) prefetch(q0 + NW * kQW);  // in flight during the gather phase
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- gather phase: 8 lanes x 4 channels per query, 8 queries per wave ----
    if (dbg != 4) {
      const int g = lane >> 3, cg = lane & 7;
      const int q = q0 + g;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* sl = slab + cg * 4;
#pragma unroll 4
      for (int pt = 0; pt < LP; ++pt) {
        const uint2 o = offs[g * LP + pt];
        const float4 ww = wts[g * LP + pt];
        const float4 v0 = *reinterpret_cast<const float4*>(sl + (o.x & 0xffffu) * D);
        const float4 v1 = *reinterpret_cast<const float4*>(sl + (o.x >> 16) * D);
        const float4 v2 = *reinterpret_cast<const float4*>(sl + (o.y & 0xffffu) * D);
        const float4 v3 = *reinterpret_cast<const float4*>(sl + (o.y >> 16) * D);
        acc.x += ww.x * v0.x + ww.y * v1.x + ww.z * v2.x + ww.w * v3.x;
        acc.y += ww.x * v0.y + ww.y * v1.y + ww.z * v2.y + ww.w * v3.y;
        acc.z += ww.x * v0.z + ww.y * v1.z + ww.z * v2.z + ww.w * v3.z;
        acc.w += ww.x * v0.w + ww.y * v1.w + ww.z * v2.w + ww.w * v3.w;
      }
      if (q < qend)
        *reinterpret_cast<float4*>(out + (((long long)b * Lq + q) * M + m) * D + cg * 4) = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// -------------------------------------------------------------------------------------------------
// LDS-staged backward, D == 32, fp32: two kernels, no floating-point atomics anywhere.
//
// Measured on MI355X (tools/ubench/lds_atomic*.hip): ds_add_f32 retires ONE LANE per 3 cycles per CU
// (192 cycles per wave instruction, independent of addresses and of the number of waves) while
// ds_add_u32 retires a whole wave instruction in 4-8 cycles.  global_atomic_add_f32 is ~14 G/s chip-wide.
// Hence:
//   msda_bwd_value_lds_d32  - grad_value.  One workgroup per (frame, head, 16-channel half).  The
//       scatter-add runs on a 32-bit FIXED-POINT accumulator slab in LDS (ds_add_u32), with one scale
//       per channel chosen from a pre-pass: scale_c = 2^30 / (max_q |grad_out[q,c]| * sum_{q,p} |w|),
//       which bounds every partial sum below 2^30.  Integer adds commute, so grad_value is bitwise
//       DETERMINISTIC (the reference's float atomicAdd is not).  Resolution: 2^-30 of the bound, i.e.
//       ~1e-6 x max|grad_out| for softmax weights at Lq = 1029.  The slab is flushed once with plain
//       coalesced stores; grad_value needs no zero-fill.  `value` is not read at all.
//   msda_bwd_locw_lds_d32   - grad_sampling_loc, grad_attn_weight.  Forward-shaped: value slab staged
//       in LDS, coordinate phase -> per-wave scratch, gather phase computes the 4 tap dot products with
//       grad_out, reduces over the 8 channel-lanes with xor shuffles and stores plainly.
// -------------------------------------------------------------------------------------------------
constexpr int kBwdVThreads = 512;

// One pass over all (query, point) samples of (b, m).  4 lanes (cg = 0..3) serve one query; 16 queries per
// wave iteration, `mult` apart.  PASS 0: scatter |a * tapweight| (rounded UP, fixed point) into wsum[row]
// -> an upper bound W_r of everything that can ever be added to a row.  PASS 1: scatter the gradient itself
// in fixed point with the per-row scale 2^30 / W_r (and grad_out normalised per channel to [-1, 1]).
template <int PASS, bool WIN = false>
__device__ __forceinline__ void bwd_value_pass(const float* __restrict__ gout, const int64_t* __restrict__ shapes,
                                               const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                                               const float* __restrict__ aw, int b, int m, int half, int S, int M,
                                               int L, int Lq, int P, int mult, int wave, int lane, float wscale,
                                               const float (&inv_mx)[4], int* __restrict__ acc,
                                               int* __restrict__ wsum, int row0 = 0, int lvl_lo = 0, int lvl_hi = 1 << 30) {
  // S: number of value rows this workgroup accumulates = index of its dummy sink row; row0: first of them in the flattened
  // pyramid (a window of a pyramid that does not fit the LDS, e.g. 64^2 + 32^2 + 16^2 cells at 512 x 512 inputs); only the
  // points of the levels lvl_lo .. lvl_hi - 1 can touch the window
  constexpr int D = 32, HD = 16, NW = kBwdVThreads / 64;
  const int LP = L * P;
  const int cg = lane & 3, slot = lane >> 2, rot = slot & 3;
  const float* rowscale = reinterpret_cast<const float*>(wsum);
  const int chunks = (Lq + 15) / 16;
  for (int ci = wave; ci < chunks; ci += NW) {
    const int n = ci * 16 + slot;
    const bool live = n < Lq;
    const int q = live ? (int)(((long long)n * mult) % Lq) : 0;
    const long long qm = ((long long)b * Lq + q) * M + m;
    float tgr[4] = {0.f, 0.f, 0.f, 0.f};
    int coff[4] = {0, 1, 2, 3};
    if constexpr (PASS == 1) {
      const float4 tg = *reinterpret_cast<const float4*>(gout + qm * D + half * HD + cg * 4);
      // normalised grad_out, channels in this lane's rotated walk order (spreads the LDS banks)
      const float tgs[4] = {tg.x * inv_mx[0], tg.y * inv_mx[1], tg.z * inv_mx[2], tg.w * inv_mx[3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        coff[j] = (j + rot) & 3;
        tgr[j] = tgs[0];
#pragma unroll
        for (int c = 1; c < 4; ++c)
          if (coff[j] == c) tgr[j] = tgs[c];
      }
    }
    // The 4 channel lanes of a query share the sample geometry: lane cg computes the points cg, cg+4, cg+8, .. (tap
    // rows, and the fixed-point factors wt * scale) ONCE and the quad reads them by DPP broadcasts, instead of every lane
    // recomputing all L*P points (the kernel is VALU-bound: 104 M VALU instructions per launch before this, PMC).
    constexpr int J = kMaxLP / 4;
    int my_r[J][4], my_ok[J];
    float my_v[J][4];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int pt = cg + 4 * j;
      my_ok[j] = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) { my_r[j][k] = S; my_v[j][k] = 0.f; }
      if (pt < LP && live && (!WIN || (pt / P >= lvl_lo && pt / P < lvl_hi))) {
        const float2 xy = *reinterpret_cast<const float2*>(loc + (qm * LP + pt) * 2);
        const float a = aw[qm * LP + pt];
        const int l = pt / P;
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)lsi[l] - row0;
        const float h_im = xy.y * H - 0.5f, w_im = xy.x * W - 0.5f;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
          const float hf = floorf(h_im), wf = floorf(w_im);
          const int h0 = (int)hf, w0 = (int)wf;
          const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
          const bool t_ok = h0 >= 0, b_ok = h0 + 1 <= H - 1, l_ok = w0 >= 0, r_ok = w0 + 1 <= W - 1;
          const int base = st + h0 * W + w0;
          int r[4] = {(t_ok && l_ok) ? base : S, (t_ok && r_ok) ? base + 1 : S, (b_ok && l_ok) ? base + W : S,
                      (b_ok && r_ok) ? base + W + 1 : S};  // row S is a dummy sink for out-of-range taps
          if constexpr (WIN) {
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = (r[k] < 0 || r[k] > S) ? S : r[k];  // ... and for taps outside this window
          }
          const float wt[4] = {hh * hw * a, hh * lw * a, lh * hw * a, lh * lw * a};
          my_ok[j] = 1;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            my_r[j][k] = r[k];
            my_v[j][k] = PASS == 0 ? fabsf(wt[k]) * wscale : wt[k] * rowscale[r[k]];
          }
        }
      }
    }
    if constexpr (PASS == 0) {
      // every lane scatters the taps of ITS points (the same 48 adds per query as one tap of every point per lane)
#pragma unroll
      for (int j = 0; j < J; ++j)
        if (my_ok[j]) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            __hip_atomic_fetch_add(wsum + my_r[j][k], __float2int_ru(my_v[j][k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
#pragma unroll
      for (int pt = 0; pt < kMaxLP; ++pt) {
        if (pt < LP) {
          constexpr int kQuad[4] = {0x00, 0x55, 0xAA, 0xFF};  // DPP quad_perm broadcasts of lane 0..3 of the quad
          const int j = pt >> 2;
          auto bc = [&](int v) {
            switch (pt & 3) {
              case 0: return __builtin_amdgcn_update_dpp(0, v, kQuad[0], 0xf, 0xf, true);
              case 1: return __builtin_amdgcn_update_dpp(0, v, kQuad[1], 0xf, 0xf, true);
              case 2: return __builtin_amdgcn_update_dpp(0, v, kQuad[2], 0xf, 0xf, true);
              default: return __builtin_amdgcn_update_dpp(0, v, kQuad[3], 0xf, 0xf, true);
            }
          };
          // (the broadcasts run for the whole wave: DPP sources must be active lanes; `ok` is uniform within a quad)
          const int ok = bc(my_ok[j]);
          int r[4];
          float ws[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            r[k] = bc(my_r[j][k]);
            ws[k] = __int_as_float(bc(__float_as_int(my_v[j][k])));
          }
          if (ok) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              int* gp = acc + r[k] * HD + cg * 4;
#pragma unroll
              for (int jj = 0; jj < 4; ++jj)
                __hip_atomic_fetch_add(gp + coff[jj], __float2int_rn(ws[k] * tgr[jj]), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        }
      }
    }
  }
}

template <bool WIN>
__global__ void __launch_bounds__(kBwdVThreads)
msda_bwd_value_lds_d32(const float* __restrict__ gout, const int64_t* __restrict__ shapes,
                       const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ aw,
                       int B, int S_all, int M, int L, int Lq, int P, int mult, float* __restrict__ gvalue, int n_parts,
                       int rows_per_part, unsigned long long* __restrict__ ts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  combo_ts_begin(ts);
  constexpr int D = 32, HD = 16, NW = kBwdVThreads / 64;
  // window of the flattened pyramid this workgroup accumulates (n_parts == 1: all of it)
  const int part = WIN ? xcd_remap(blockIdx.x, gridDim.x) / 2 % n_parts : 0;
  const int row0 = part * rows_per_part;
  const int S = WIN ? min(rows_per_part, S_all - row0) : S_all;
  int lvl_lo = WIN ? L : 0, lvl_hi = WIN ? 0 : L;
  if constexpr (WIN) {
    for (int l = 0; l < L; ++l) {
      const int st = (int)lsi[l], en = st + (int)(shapes[2 * l] * shapes[2 * l + 1]);
      if (st < row0 + S && en > row0) { lvl_lo = min(lvl_lo, l); lvl_hi = max(lvl_hi, l + 1); }
    }
  }
  int* acc = reinterpret_cast<int*>(smem);                         // [S+1][16] fixed point
  int* wsum = acc + (S + 1) * HD;                                  // [S+1] weight bound -> row scale
  float* red = reinterpret_cast<float*>(wsum + ((S + 1 + 3) & ~3));  // [NW][20] block-reduction scratch
  const int LP = L * P;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int half = logical & 1;
  const int bm = (logical >> 1) / n_parts;
  const int m = bm % M, b = bm / M;
  const int cg = tid & 3;  // this lane's 4 channels: half*16 + cg*4 ..

  for (int r = tid >> 2; r <= S; r += kBwdVThreads / 4)
    *reinterpret_cast<int4*>(acc + r * HD + cg * 4) = make_int4(0, 0, 0, 0);
  for (int r = tid; r <= S; r += kBwdVThreads) wsum[r] = 0;

  // ---- pre-pass: per-channel max |grad_out| over the queries of (b, m), and sum |attention weight| ----
  float mx[4] = {0.f, 0.f, 0.f, 0.f};
  float sa = 0.f;
  bool nan = false;
  for (int q = tid >> 2; q < Lq; q += kBwdVThreads / 4) {
    const float4 t = *reinterpret_cast<const float4*>(gout + (((long long)b * Lq + q) * M + m) * D + half * HD + cg * 4);
    mx[0] = fmaxf(mx[0], fabsf(t.x)); mx[1] = fmaxf(mx[1], fabsf(t.y));
    mx[2] = fmaxf(mx[2], fabsf(t.z)); mx[3] = fmaxf(mx[3], fabsf(t.w));
    nan |= !(t.x == t.x) || !(t.y == t.y) || !(t.z == t.z) || !(t.w == t.w);  // fmaxf drops NaNs
  }
  for (int i = tid; i < Lq * LP; i += kBwdVThreads) {
    const int q = i / LP, pt = i - q * LP;
    if (!WIN || (pt / P >= lvl_lo && pt / P < lvl_hi)) sa += fabsf(aw[(((long long)b * Lq + q) * M + m) * LP + pt]);
  }
#pragma unroll
  for (int s = 4; s < 64; s <<= 1)  // lanes with equal cg
#pragma unroll
    for (int c = 0; c < 4; ++c) mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], s));
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) sa += __shfl_xor(sa, s);
  const bool wave_nan = __any(nan);
  if (lane < 4) {
#pragma unroll
    for (int c = 0; c < 4; ++c) red[wave * 20 + lane * 4 + c] = mx[c];
  }
  if (lane == 0) { red[wave * 20 + 16] = sa; red[wave * 20 + 17] = wave_nan ? 1.f : 0.f; }
  __syncthreads();
  float mxc[4], inv_mx[4];
  float tot = 0.f, bad = 0.f;
  for (int w = 0; w < NW; ++w) { tot += red[w * 20 + 16]; bad += red[w * 20 + 17]; }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float mm = 0.f;
    for (int w = 0; w < NW; ++w) mm = fmaxf(mm, red[w * 20 + cg * 4 + c]);
    if (bad != 0.f || !(mm < 3.0e38f) || !(tot < 3.0e38f)) mm = __builtin_nanf("");
    mxc[c] = mm;
    inv_mx[c] = (mm > 0.f) ? 1.f / mm : (mm == mm ? 0.f : mm);
  }
  // weights are accumulated with scale 2^30 / sum|a| (every row sum <= sum|a|)
  const float wscale = tot > 0.f ? 1073741824.f / tot : 0.f;
  const float inv_wscale = tot > 0.f ? tot * (1.f / 1073741824.f) : 0.f;

  float dummy[4] = {0.f, 0.f, 0.f, 0.f};
  bwd_value_pass<0, WIN>(gout, shapes, lsi, loc, aw, b, m, half, S, M, L, Lq, P, mult, wave, lane, wscale, dummy, acc, wsum, row0,
                    lvl_lo, lvl_hi);
  __syncthreads();
  // W_r (upper bound, rounded up) -> per-row fixed-point scale 2^30 / W_r, stored in place as float
  for (int r = tid; r <= S; r += kBwdVThreads) {
    const float wr = (float)wsum[r] * inv_wscale * 1.0001f;
    reinterpret_cast<float*>(wsum)[r] = wr > 0.f ? 1073741824.f / wr : 0.f;
  }
  __syncthreads();
  bwd_value_pass<1, WIN>(gout, shapes, lsi, loc, aw, b, m, half, S, M, L, Lq, P, mult, wave, lane, wscale, inv_mx, acc, wsum, row0,
                    lvl_lo, lvl_hi);
  __syncthreads();
  {
    float* gb = gvalue + (((long long)b * S_all + row0) * M + m) * D + half * HD + cg * 4;
    const float* rowscale = reinterpret_cast<const float*>(wsum);
    for (int r = tid >> 2; r < S; r += kBwdVThreads / 4) {
      const int4 v = *reinterpret_cast<const int4*>(acc + r * HD + cg * 4);
      const float rs = rowscale[r];
      const float f = rs > 0.f ? 1.f / rs : 0.f;
      *reinterpret_cast<float4*>(gb + (long long)r * M * D) =
          make_float4((float)v.x * f * mxc[0], (float)v.y * f * mxc[1], (float)v.z * f * mxc[2], (float)v.w * f * mxc[3]);
    }
  }
  combo_ts_end(ts);
}

__global__ void __launch_bounds__(768)
msda_bwd_locw_lds_d32(const float* __restrict__ gout, const float* __restrict__ value,
                      const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                      const float* __restrict__ loc, const float* __restrict__ aw, int B, int S, int M, int L, int Lq,
                      int P, int QT, float* __restrict__ gloc, float* __restrict__ gaw, unsigned long long* __restrict__ ts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  combo_ts_begin(ts);
  constexpr int D = 32;
  float* slab = reinterpret_cast<float*>(smem);
  const int LP = L * P;
  const int NW = blockDim.x >> 6;
  const int slab_bytes = (S + 1) * D * 4;
  float4* par_all = reinterpret_cast<float4*>(smem + slab_bytes);  // (lh, lw, a*W, a*H)
  uint2* offs_all = reinterpret_cast<uint2*>(smem + slab_bytes + NW * kQW * LP * 16);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = logical % QT;
  const int bm = logical / QT;
  const int m = bm % M, b = bm / M;

  {
    const float* vb = value + ((long long)b * S * M + m) * D + (lane & 7) * 4;
    for (int r0 = wave * 8; r0 < S; r0 += NW * 8) {
      const int r = r0 + (lane >> 3);
      if (r < S) dma16(vb + (long long)r * M * D, slab + r0 * D);
    }
    if (tid < 8) *reinterpret_cast<float4*>(slab + S * D + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  int lvH[kMaxLevels], lvW[kMaxLevels], lvS[kMaxLevels];
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l) {
    lvH[l] = l < L ? (int)shapes[2 * l] : 1;
    lvW[l] = l < L ? (int)shapes[2 * l + 1] : 1;
    lvS[l] = l < L ? (int)lsi[l] : 0;
  }
  float4* par = par_all + wave * kQW * LP;
  uint2* offs = offs_all + wave * kQW * LP;
  const int qbeg = (int)(((long long)Lq * qt) / QT), qend = (int)(((long long)Lq * (qt + 1)) / QT);
  const int npairs = kQW * LP;
  __syncthreads();

  for (int q0 = qbeg + wave * kQW; q0 < qend; q0 += NW * kQW) {
    for (int i = lane; i < npairs; i += 64) {
      const int ql = i / LP, pt = i - ql * LP;
      const int q = q0 + ql;
      float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
      unsigned r01 = (unsigned)S | ((unsigned)S << 16), r23 = r01;
      if (q < qend) {
        const long long e = (((long long)b * Lq + q) * M + m) * LP + pt;
        const float2 xy = *reinterpret_cast<const float2*>(loc + e * 2);
        const float a = aw[e];
        const int l = pt / P;
        int H = lvH[0], W = lvW[0], st = lvS[0];
#pragma unroll
        for (int k = 1; k < kMaxLevels; ++k)
          if (l == k) { H = lvH[k]; W = lvW[k]; st = lvS[k]; }
        const float h_im = xy.y * H - 0.5f, w_im = xy.x * W - 0.5f;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
          const float hf = floorf(h_im), wf = floorf(w_im);
          const int h0 = (int)hf, w0 = (int)wf;
          const bool t_ok = h0 >= 0, b_ok = h0 + 1 <= H - 1, l_ok = w0 >= 0, r_ok = w0 + 1 <= W - 1;
          const int base = st + h0 * W + w0;
          const unsigned r0 = (t_ok && l_ok) ? (unsigned)base : (unsigned)S;
          const unsigned r1 = (t_ok && r_ok) ? (unsigned)(base + 1) : (unsigned)S;
          const unsigned r2 = (b_ok && l_ok) ? (unsigned)(base + W) : (unsigned)S;
          const unsigned r3 = (b_ok && r_ok) ? (unsigned)(base + W + 1) : (unsigned)S;
          r01 = r0 | (r1 << 16);
          r23 = r2 | (r3 << 16);
          pv = make_float4(h_im - hf, w_im - wf, a * W, a * H);
        }
      }
      par[i] = pv;
      offs[i] = make_uint2(r01, r23);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      const int g = lane >> 3, cg = lane & 7;
      const int q = q0 + g;
      const int qc = q < qend ? q : qbeg;
      const long long qm = ((long long)b * Lq + qc) * M + m;
      const float4 tg = *reinterpret_cast<const float4*>(gout + qm * D + cg * 4);
      const float* sl = slab + cg * 4;
      float keep_w[2] = {0.f, 0.f}, keep_x[2] = {0.f, 0.f}, keep_y[2] = {0.f, 0.f};
#pragma unroll 2
      for (int pt = 0; pt < LP; ++pt) {
        const uint2 o = offs[g * LP + pt];
        const float4 pp = par[g * LP + pt];
        const float4 v0 = *reinterpret_cast<const float4*>(sl + (o.x & 0xffffu) * D);
        const float4 v1 = *reinterpret_cast<const float4*>(sl + (o.x >> 16) * D);
        const float4 v2 = *reinterpret_cast<const float4*>(sl + (o.y & 0xffffu) * D);
        const float4 v3 = *reinterpret_cast<const float4*>(sl + (o.y >> 16) * D);
        const float d0 = tg.x * v0.x + tg.y * v0.y + tg.z * v0.z + tg.w * v0.w;
        const float d1 = tg.x * v1.x + tg.y * v1.y + tg.z * v1.z + tg.w * v1.w;
        const float d2 = tg.x * v2.x + tg.y * v2.y + tg.z * v2.z + tg.w * v2.w;
        const float d3 = tg.x * v3.x + tg.y * v3.y + tg.z * v3.z + tg.w * v3.w;
        const float lh = pp.x, lw = pp.y, hh = 1.f - lh, hw = 1.f - lw;
        float sw = hh * hw * d0 + hh * lw * d1 + lh * hw * d2 + lh * lw * d3;  // d out / d w
        float sy = (-hw * d0 - lw * d1 + hw * d2 + lw * d3) * pp.w;              // * a * H
        float sx = (-hh * d0 + hh * d1 - lh * d2 + lh * d3) * pp.z;              // * a * W
        // sum over the 8 channel lanes of the query with DPP adds (quad swaps, then the mirror of the 8-lane half row);
        // `__shfl_xor` went through the LDS pipe (ds_bpermute_b32) three dependent times per point
        sw += dpp_f<0xB1>(sw); sx += dpp_f<0xB1>(sx); sy += dpp_f<0xB1>(sy);     // quad_perm [1,0,3,2]
        sw += dpp_f<0x4E>(sw); sx += dpp_f<0x4E>(sx); sy += dpp_f<0x4E>(sy);     // quad_perm [2,3,0,1]
        sw += dpp_f<0x141>(sw); sx += dpp_f<0x141>(sx); sy += dpp_f<0x141>(sy);  // row_half_mirror
        // lane cg keeps points cg and cg+8 -> coalesced stores below
        if ((pt & 7) == cg) {
          if (pt < 8) { keep_w[0] = sw; keep_x[0] = sx; keep_y[0] = sy; }
          else { keep_w[1] = sw; keep_x[1] = sx; keep_y[1] = sy; }
        }
      }
      if (q < qend) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pt = cg + 8 * h;
          if (pt < LP) {
            gaw[qm * LP + pt] = keep_w[h];
            *reinterpret_cast<float2*>(gloc + (qm * LP + pt) * 2) = make_float2(keep_x[h], keep_y[h]);
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  combo_ts_end(ts);
}

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  const long long cap = 256LL * 16;  // 256 CUs x 16 blocks, grid-stride the rest
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// -------------------------------------------------------------------------------------------------
// LDS-staged forward v2 ("tap-parallel, persistent"), D == 32, fp32, L*P == LPc.
//
// Two changes against msda_fwd_lds_d32, both from measurements of that kernel (gather phase LDS-bound with 29 %
// bank-conflict cycles; slab staged QT = 4 times per (frame, head), 26 us of the 72 us):
//  * persistent workgroups: grid = #CUs, each owns a CONTIGUOUS range of (frame, head, query-tile) tiles and
//    re-stages the slab only when (frame, head) changes: 2 stagings per CU instead of 5 at bs = 8.
//  * conflict-free gather: a ds_read_b128 is served in four fixed 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,
//    28-31} and the same +32); a 128-B value row covers half of the 64 banks, selected by the row's parity.  v1 gave
//    each 8-lane group its own random row, so two rows of equal parity collided in 3 of 4 accesses.  Here the 16 lanes
//    of a hardware group read the two HORIZONTALLY ADJACENT rows of one bilinear tap pair (r, r+1: opposite parity,
//    both 64-B halves) = all 64 banks exactly once.  32 lanes serve one (query, point): 4 taps x 2 halves x 4 lanes
//    x 16 B; a lane accumulates ITS tap over the L*P points and the four taps meet at the end in two DPP row
//    rotations.  Per-(query, point, tap) {row byte offset, weight} pairs come from a per-wave LDS scratch that a
//    coordinate phase (one lane per (query, point)) fills, as in v1.
// -------------------------------------------------------------------------------------------------
constexpr int kTapQW = 4;  // queries per wave iteration (2 gather steps of 2 queries); small -> 16 waves fit beside the slab

typedef float v2f __attribute__((ext_vector_type(2)));


template <int LPc>
__global__ void __launch_bounds__(1024)
msda_fwd_tap_d32(const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                 const float* __restrict__ loc, const float* __restrict__ aw, int B, int S, int M, int L, int Lq, int P,
                 int QT, float* __restrict__ out, int dbg, unsigned long long* __restrict__ ts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = 32, QW = kTapQW, PRE = (QW * LPc + 63) / 64;
  // ts != nullptr: device-side timing of this launch (works inside a replayed hipGraph, where HIP refuses event records):
  // ts = {min start, workgroups done, sum of durations, launches} in wall-clock ticks, see combo_msda_set_timing_buffer
  if (ts && threadIdx.x == 0) atomicMin(&ts[0], (unsigned long long)wall_clock64());
  float* slab = reinterpret_cast<float*>(smem);
  const int NW = blockDim.x >> 6;
  const unsigned slab_bytes = (unsigned)(S + 1) * D * 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* ent = smem + slab_bytes + wave * (QW * LPc * 32);  // [QW][LPc][4 taps] x {u32 row byte offset, f32 weight}

  // gather roles (see the header comment): lane -> (query of the pair, 64-B half, tap, 16-B piece)
  const int qsel = lane >> 5, half = (lane >> 4) & 1, kq = (lane >> 2) & 3, c4 = lane & 3;
  const int tap = ((half ? 0x3102 : 0x1320) >> (kq * 4)) & 3;
  const unsigned lane_off = (unsigned)(half * 64 + c4 * 16);
  const unsigned ent_lane = (unsigned)(tap * (QW * LPc * 8) + qsel * LPc * 8);

  int lvH[kMaxLevels], lvW[kMaxLevels], lvS[kMaxLevels];
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l) {
    lvH[l] = l < L ? (int)shapes[2 * l] : 1;
    lvW[l] = l < L ? (int)shapes[2 * l + 1] : 1;
    lvS[l] = l < L ? (int)lsi[l] : 0;
  }
  // coordinate-phase constants: lane serves pairs i = lane + 64 j -> (query slot, point, level geometry)
  int c_ql[PRE], c_H[PRE], c_W[PRE], c_st[PRE];
  long long c_eoff[PRE];
  bool c_on[PRE];
#pragma unroll
  for (int j = 0; j < PRE; ++j) {
    const int i = lane + j * 64;
    const int ql = i / LPc, pt = i - ql * LPc;
    const int l = pt / P;
    c_on[j] = i < QW * LPc;
    c_ql[j] = ql;
    c_eoff[j] = (long long)ql * M * LPc + pt;
    int H = lvH[0], W = lvW[0], st = lvS[0];
#pragma unroll
    for (int k = 1; k < kMaxLevels; ++k)
      if (l == k) { H = lvH[k]; W = lvW[k]; st = lvS[k]; }
    c_H[j] = H; c_W[j] = W; c_st[j] = st;
  }

  // Work list of this workgroup (NS = B*M slabs, G workgroups):
  //   phase 1: n_full = NS / G whole slabs (all Lq queries), slab = it*G + wg: the 8 heads of a frame run on
  //            neighbouring CUs of one XCD at the same time (one HBM fetch per value row);
  //   phase 2: the R = NS mod G left-over slabs are cut into QT query tiles each and the R*QT tiles are dealt out
  //            contiguously, so the (G / R) workgroups sharing a slab stage it at the same time from the same L2.
  const int NS = B * M, G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);
  const int n_full = NS / G, R = NS - n_full * G;
  const int p_beg = (int)(((long long)R * QT * wg) / G), p_end = (int)(((long long)R * QT * (wg + 1)) / G);
  const int n_items = n_full + (p_end - p_beg);
  int cur_bm = -1;
  for (int it = 0; it < n_items; ++it) {
    int bm, qbeg = 0, qend = Lq;
    if (it < n_full) {
      bm = it * G + wg;
    } else {
      const int p = p_beg + (it - n_full);
      const int qt = p % QT;
      bm = n_full * G + p / QT;
      qbeg = (int)(((long long)Lq * qt) / QT);
      qend = (int)(((long long)Lq * (qt + 1)) / QT);
    }
    const int m = bm % M, b = bm / M;
    float2 pxy[PRE];
    float pa[PRE];
    auto prefetch = [&](int q0) {
      const long long base = (((long long)b * Lq + q0) * M + m) * LPc;
#pragma unroll
      for (int j = 0; j < PRE; ++j) {
        pxy[j] = make_float2(-8.f, -8.f);
        pa[j] = 0.f;
        if (c_on[j] && q0 + c_ql[j] < qend) {
          const long long e = base + c_eoff[j];
          pxy[j] = *reinterpret_cast<const float2*>(loc + e * 2);
          pa[j] = aw[e];
        }
      }
    };
    int q0 = qbeg + wave * QW;
    if (bm != cur_bm) {  // (uniform over the workgroup)
      if (cur_bm >= 0) __syncthreads();  // every wave is done gathering from the previous slab
      const float* vb = value + ((long long)b * S * M + m) * D + (lane & 7) * 4;
      for (int r0 = wave * 8; r0 < S; r0 += NW * 8) {  // LDS-DMA, 8 rows (1 KiB) per wave instruction
        const int r = r0 + (lane >> 3);
        if (r < S) dma16(vb + (long long)r * M * D, slab + r0 * D);
      }
      if (tid < 8) *reinterpret_cast<float4*>(slab + S * D + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q0 < qend) prefetch(q0);
      __syncthreads();  // drains the DMA and publishes the slab
      cur_bm = bm;
    } else if (q0 < qend) {
      prefetch(q0);
    }

    if (dbg == 1) continue;  // ablation: staging only
    for (; q0 < qend; q0 += NW * QW) {
      // ---- coordinate phase ----
#pragma unroll
      for (int j = 0; j < PRE; ++j) {
        if (c_on[j] && dbg != 3) {
          const unsigned zrow = (unsigned)S * 128u;
          unsigned a0 = zrow, a1 = zrow, a2 = zrow, a3 = zrow;
          float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
          const float2 xy = pxy[j];
          const float a = pa[j];
          const int H = c_H[j], W = c_W[j], st = c_st[j];
          const float h_im = xy.y * H - 0.5f, w_im = xy.x * W - 0.5f;
          if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h0 = (int)hf, w0i = (int)wf;
            const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
            const bool t_ok = h0 >= 0, b_ok = h0 + 1 <= H - 1, l_ok = w0i >= 0, r_ok = w0i + 1 <= W - 1;
            const unsigned base = (unsigned)(st + h0 * W + w0i) * 128u;
            if (t_ok && l_ok) a0 = base;
            if (t_ok && r_ok) a1 = base + 128u;
            if (b_ok && l_ok) a2 = base + (unsigned)W * 128u;
            if (b_ok && r_ok) a3 = base + (unsigned)W * 128u + 128u;
            w0 = hh * hw * a; w1 = hh * lw * a; w2 = lh * hw * a; w3 = lh * lw * a;
          }
          // layout [tap][query][point] x 8 B: consecutive lanes write consecutive 8-B entries (conflict-free
          // ds_write_b64); a gather lane reads two points of ITS tap per ds_read_b128
          uint2* e = reinterpret_cast<uint2*>(ent + (lane + j * 64) * 8);
          e[0] = make_uint2(a0, __float_as_uint(w0));
          e[QW * LPc] = make_uint2(a1, __float_as_uint(w1));
          e[2 * QW * LPc] = make_uint2(a2, __float_as_uint(w2));
          e[3 * QW * LPc] = make_uint2(a3, __float_as_uint(w3));
        }
      }
      if (q0 + NW * QW < qend) prefetch(q0 + NW * QW);  // in flight during the gather phase
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // ---- gather phase: 2 queries per wave instruction ----
      if (dbg != 4)
#pragma unroll 2
      for (int s = 0; s < QW / 2; ++s) {
        const char* e = ent + ent_lane + s * (2 * LPc * 8);
        uint4 en[LPc / 2];
#pragma unroll
        for (int pp = 0; pp < LPc / 2; ++pp) en[pp] = *reinterpret_cast<const uint4*>(e + pp * 16);
        v2f acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
#pragma unroll
        for (int pp = 0; pp < LPc / 2; ++pp) {
          const float4 va = *reinterpret_cast<const float4*>(smem + en[pp].x + lane_off);
          const float4 vb = *reinterpret_cast<const float4*>(smem + en[pp].z + lane_off);
          const float wa = __uint_as_float(en[pp].y), wb = __uint_as_float(en[pp].w);
          acc01 = __builtin_elementwise_fma((v2f){wa, wa}, (v2f){va.x, va.y}, acc01);
          acc23 = __builtin_elementwise_fma((v2f){wa, wa}, (v2f){va.z, va.w}, acc23);
          acc01 = __builtin_elementwise_fma((v2f){wb, wb}, (v2f){vb.x, vb.y}, acc01);
          acc23 = __builtin_elementwise_fma((v2f){wb, wb}, (v2f){vb.z, vb.w}, acc23);
        }
        float4 acc = make_float4(acc01.x, acc01.y, acc23.x, acc23.y);
        // the four taps of a (query, half, piece) sit in the four quads of a 16-lane row
        acc.x += dpp_f<0x128>(acc.x); acc.y += dpp_f<0x128>(acc.y); acc.z += dpp_f<0x128>(acc.z); acc.w += dpp_f<0x128>(acc.w);
        acc.x += dpp_f<0x124>(acc.x); acc.y += dpp_f<0x124>(acc.y); acc.z += dpp_f<0x124>(acc.z); acc.w += dpp_f<0x124>(acc.w);
        const int q = q0 + 2 * s + qsel;
        if (kq == 0 && q < qend)
          *reinterpret_cast<float4*>(out + (((long long)b * Lq + q) * M + m) * D + half * 16 + c4 * 4) = acc;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (ts) {  // the last workgroup to finish closes the measurement: duration = its end - the earliest start
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long t1 = wall_clock64();
      __threadfence();
      if (atomicAdd(&ts[1], 1ull) == (unsigned long long)gridDim.x - 1ull) {
        const unsigned long long t0 = atomicExch(&ts[0], ~0ull);  // (also re-arms the slot for the next replay)
        atomicExch(&ts[1], 0ull);
        atomicAdd(&ts[2], t1 - t0);
        atomicAdd(&ts[3], 1ull);
      }
    }
  }
}

inline bool check_common(int B, int S, int M, int D, int L, int Lq, int P) {
  return B > 0 && S > 0 && M > 0 && D > 0 && L > 0 && L <= kMaxLevels && Lq > 0 && P > 0;
}

size_t fwd_lds_bytes(int S, int L, int P, int nw) { return (size_t)(S + 1) * 128 + (size_t)nw * kQW * L * P * 24; }
size_t fwd_tap_lds_bytes(int S, int LP, int nw) { return (size_t)(S + 1) * 128 + (size_t)nw * kTapQW * LP * 32; }
size_t bwd_value_lds_bytes(int S) {
  return (size_t)(S + 1) * 64 + (size_t)((S + 1 + 3) & ~3) * 4 + (size_t)(kBwdVThreads / 64) * 20 * 4;
}
constexpr size_t kLdsLimit = 160 * 1024;


template <typename T>
int msda_forward(const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc, const T* aw, int B, int S,
                 int M, int D, int L, int Lq, int P, T* out, int algo, hipStream_t stream) {
  if (!value || !shapes || !lsi || !loc || !aw || !out || !check_common(B, S, M, D, L, Lq, P)) return COMBO_EINVAL;
  if ((long long)B * Lq * M * D >= (1LL << 31) * 8) return COMBO_EINVAL;
  bool lds_ok = false;
  if constexpr (sizeof(T) == 4)
    lds_ok = (D == 32) && S < 65535 && L * P <= kMaxLP && fwd_lds_bytes(S, L, P, 8) <= kLdsLimit;
  if (algo == 2 && !lds_ok) return COMBO_EINVAL;
  bool tap_ok = false;
  if constexpr (sizeof(T) == 4)
    tap_ok = (D == 32) && S < 65535 && (L * P == 12 || L * P == 16) && fwd_tap_lds_bytes(S, L * P, 4) <= kLdsLimit;
  if (algo == 3 && !tap_ok) return COMBO_EINVAL;
  if constexpr (sizeof(T) == 4) {
    if (tap_ok && (algo == 0 || algo == 3)) {
      int nw = 16;
      while (nw > 4 && fwd_tap_lds_bytes(S, L * P, nw) > kLdsLimit) --nw;
      static int n_cu = 0;
      if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return COMBO_EINVAL;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
      }
      // left-over slabs (B*M mod #CUs) are cut into QT tiles so that their tiles divide evenly over the CUs
      const int NS = B * M, rem = NS % n_cu;
      auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
      int QT = rem ? n_cu / gcd(rem, n_cu) : 1;
      const int qt_max = Lq / (nw * kTapQW) > 0 ? Lq / (nw * kTapQW) : 1;
      if (QT > qt_max) QT = qt_max;
      const int grid = NS >= n_cu ? n_cu : (NS * QT < n_cu ? NS * QT : n_cu);
      const size_t lds = fwd_tap_lds_bytes(S, L * P, nw);
      const int dbg = 0;  // (the kernel's ablation bits - 1: staging only, 3: no coordinate phase, 4: no gather - are compiled in)
      static ComboDevFlag attr12, attr16;
      if (L * P == 12) {
        if (!attr12.is_set()) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_fwd_tap_d32<12>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
          if (e != hipSuccess) return (int)e;
          attr12.mark();
        }
        hipLaunchKernelGGL(msda_fwd_tap_d32<12>, dim3(grid), dim3(nw * 64), lds, stream, value, shapes, lsi, loc, aw, B,
                           S, M, L, Lq, P, QT, out, dbg, combo_timing_next_slot(COMBO_TS_MSDA_FWD, (double)B * ((double)(S + Lq) * M * D + 3.0 * Lq * M * L * P) * 4.0));
      } else {
        if (!attr16.is_set()) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_fwd_tap_d32<16>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
          if (e != hipSuccess) return (int)e;
          attr16.mark();
        }
        hipLaunchKernelGGL(msda_fwd_tap_d32<16>, dim3(grid), dim3(nw * 64), lds, stream, value, shapes, lsi, loc, aw, B,
                           S, M, L, Lq, P, QT, out, dbg, combo_timing_next_slot(COMBO_TS_MSDA_FWD, (double)B * ((double)(S + Lq) * M * D + 3.0 * Lq * M * L * P) * 4.0));
      }
      return (int)hipGetLastError();
    }
    if (lds_ok && algo != 1) {
      // enough workgroups to fill 256 CUs a few times; each re-stages the slab from L2 (cheap)
      const int nw = fwd_lds_bytes(S, L, P, 12) <= kLdsLimit ? 12 : 8;
      int QT = 1;
      while ((long long)B * M * QT < 1024 && QT < 8 && Lq / (QT * 2) >= nw * kQW) QT *= 2;
      const size_t lds = fwd_lds_bytes(S, L, P, nw);
      static ComboDevFlag attr_set;
      if (!attr_set.is_set()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_fwd_lds_d32),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
        if (e != hipSuccess) return (int)e;
        attr_set.mark();
      }
      const int dbg = 0;
      hipLaunchKernelGGL(msda_fwd_lds_d32, dim3(B * M * QT), dim3(nw * 64), lds, stream, value, shapes, lsi, loc,
                         aw, B, S, M, L, Lq, P, QT, out, dbg);
      return (int)hipGetLastError();
    }
  }
  if (D % 4 == 0) {
    const long long total = (long long)B * Lq * M * (D / 4);
    hipLaunchKernelGGL((msda_fwd_generic<T, 4>), dim3(grid_for(total, 256)), dim3(256), 0, stream, value, shapes, lsi,
                       loc, aw, B, S, M, D, L, Lq, P, out, total);
  } else {
    const long long total = (long long)B * Lq * M * D;
    hipLaunchKernelGGL((msda_fwd_generic<T, 1>), dim3(grid_for(total, 256)), dim3(256), 0, stream, value, shapes, lsi,
                       loc, aw, B, S, M, D, L, Lq, P, out, total);
  }
  return (int)hipGetLastError();
}

template <typename T, int G, bool VALUE = true>
void launch_bwd_generic(const T* gout, const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc,
                        const T* aw, int B, int S, int M, int D, int L, int Lq, int P, T* gv, T* gl, T* gw,
                        hipStream_t stream) {
  long long total = (long long)B * Lq * M * G;
  total = (total + 255) / 256 * 256;
  hipLaunchKernelGGL((msda_bwd_generic<T, G, VALUE>), dim3(grid_for(total, 256)), dim3(256), 0, stream, gout, value, shapes,
                     lsi, loc, aw, B, S, M, D, L, Lq, P, gv, gl, gw, total);
}

// A pyramid whose value slab does not fit the LDS (S = 5376 at 512 x 512 inputs, BASELINE configs[3]): grad_value on the
// fixed-point LDS kernel over WINDOWS of <= kBwdWindowRows rows (two workgroups per CU), grad_loc / grad_w on the generic
// gather kernel with its value-gradient atomics compiled out.  (The all-in-one generic kernel, global float atomics for
// grad_value, took 62 ms per layer at BT = 80.)
constexpr int kBwdWindowRows = 1100;
inline bool bwd_windowed_ok(int S, int D, int L, int P, int elem_bytes) {
  return elem_bytes == 4 && D == 32 && S < 65535 && L * P <= kMaxLP && L <= kMaxLevels;
}

template <typename T>
int msda_backward(const T* gout, const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc, const T* aw,
                  int B, int S, int M, int D, int L, int Lq, int P, T* gv, T* gl, T* gw, int algo,
                  hipStream_t stream) {
  if (!gout || !value || !shapes || !lsi || !loc || !aw || !gv || !gl || !gw || !check_common(B, S, M, D, L, Lq, P))
    return COMBO_EINVAL;
  bool lds_ok = false;
  if constexpr (sizeof(T) == 4)
    lds_ok = (D == 32) && S < 65535 && L * P <= kMaxLP && fwd_lds_bytes(S, L, P, 8) <= kLdsLimit;
  if (algo == 2 && !lds_ok) return COMBO_EINVAL;
  if constexpr (sizeof(T) == 4) {
    if (lds_ok && algo != 1) {
      static ComboDevFlag attr_set;
      if (!attr_set.is_set()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_bwd_value_lds_d32<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_bwd_locw_lds_d32),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
        if (e != hipSuccess) return (int)e;
        attr_set.mark();
      }
      // query stride: ~Lq/16 and coprime with Lq, so the 16 queries of one wave instruction are far apart
      int mult = Lq / 16 + 1;
      auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
      while (gcd(mult, Lq) != 1) ++mult;
      hipLaunchKernelGGL(msda_bwd_value_lds_d32<false>, dim3(B * M * 2), dim3(kBwdVThreads), bwd_value_lds_bytes(S), stream,
                         gout, shapes, lsi, loc, aw, B, S, M, L, Lq, P, mult, gv, 1, S,
                         // algorithmic bytes of the PAIR of kernels (value, grad_out, loc, w once; three gradients once) on
                         // the first slot, 0 on the second: the per-kind sums give the pair's rate
                         combo_timing_next_slot(COMBO_TS_MSDA_BWD, 4.0 * B * (2.0 * ((double)S + Lq) * M * D + 6.0 * (double)Lq * M * L * P)));
      hipError_t e1 = hipGetLastError();
      if (e1 != hipSuccess) return (int)e1;
      const int nw = fwd_lds_bytes(S, L, P, 12) <= kLdsLimit ? 12 : 8;
      int QT = 1;
      while ((long long)B * M * QT < 1024 && QT < 8 && Lq / (QT * 2) >= nw * kQW) QT *= 2;
      hipLaunchKernelGGL(msda_bwd_locw_lds_d32, dim3(B * M * QT), dim3(nw * 64), fwd_lds_bytes(S, L, P, nw), stream,
                         gout, value, shapes, lsi, loc, aw, B, S, M, L, Lq, P, QT, gl, gw,
                         combo_timing_next_slot(COMBO_TS_MSDA_BWD, 0.0));
      return (int)hipGetLastError();
    }
  }
  if constexpr (sizeof(T) == 4) {
    if (bwd_windowed_ok(S, D, L, P, 4) && algo != 1) {
      static ComboDevFlag attr_w;
      if (!attr_w.is_set()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(msda_bwd_value_lds_d32<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
        if (e != hipSuccess) return (int)e;
        attr_w.mark();
      }
      const int n_parts = (S + kBwdWindowRows - 1) / kBwdWindowRows;
      const int rows = (S + n_parts - 1) / n_parts;
      int mult = Lq / 16 + 1;
      auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
      while (gcd(mult, Lq) != 1) ++mult;
      hipLaunchKernelGGL(msda_bwd_value_lds_d32<true>, dim3(B * M * 2 * n_parts), dim3(kBwdVThreads), bwd_value_lds_bytes(rows), stream,
                         gout, shapes, lsi, loc, aw, B, S, M, L, Lq, P, mult, gv, n_parts, rows, (unsigned long long*)nullptr);
      hipError_t e1 = hipGetLastError();
      if (e1 != hipSuccess) return (int)e1;
      launch_bwd_generic<T, 8, false>(gout, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, gv, gl, gw, stream);
      return (int)hipGetLastError();
    }
  }
  const int G = (D % 4 == 0) ? D / 4 : 0;
  switch (G) {
    case 1: launch_bwd_generic<T, 1>(gout, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, gv, gl, gw, stream); break;
    case 2: launch_bwd_generic<T, 2>(gout, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, gv, gl, gw, stream); break;
    case 4: launch_bwd_generic<T, 4>(gout, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, gv, gl, gw, stream); break;
    case 8: launch_bwd_generic<T, 8>(gout, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, gv, gl, gw, stream); break;
    case 16: launch_bwd_generic<T, 16>(gout, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, gv, gl, gw, stream); break;
    case 32: launch_bwd_generic<T, 32>(gout, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, gv, gl, gw, stream); break;
    case 64: launch_bwd_generic<T, 64>(gout, value, shapes, lsi, loc, aw, B, S, M, D, L, Lq, P, gv, gl, gw, stream); break;
    default: {
      const long long total = (long long)B * Lq * M * L * P;
      hipLaunchKernelGGL((msda_bwd_serial<T>), dim3(grid_for(total, 256)), dim3(256), 0, stream, gout, value, shapes,
                         lsi, loc, aw, B, S, M, D, L, Lq, P, gv, gl, gw, total);
    }
  }
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

// 0 when combo_msda_backward_* will take the LDS kernels, which write every element of the three gradients; 1 when the
// generic global-atomics kernels run and the caller has to zero-fill the outputs (mirrors the dispatch in msda_backward)
int combo_msda_backward_needs_zero(int S, int D, int L, int P, int elem_bytes, int algo) {
  const bool lds_ok = elem_bytes == 4 && D == 32 && S < 65535 && L * P <= kMaxLP && fwd_lds_bytes(S, L, P, 8) <= kLdsLimit &&
                      bwd_value_lds_bytes(S) <= kLdsLimit;
  return ((lds_ok || bwd_windowed_ok(S, D, L, P, elem_bytes)) && algo != 1) ? 0 : 1;
}

int combo_msda_forward_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                           const float* sampling_loc, const float* attn_weight, int B, int S, int M, int D, int L,
                           int Lq, int P, float* out, int algo, combo_stream_t stream) {
  return msda_forward<float>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, B, S, M, D, L, Lq, P,
                             out, algo, (hipStream_t)stream);
}
int combo_msda_forward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                           const double* sampling_loc, const double* attn_weight, int B, int S, int M, int D, int L,
                           int Lq, int P, double* out, int algo, combo_stream_t stream) {
  return msda_forward<double>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, B, S, M, D, L, Lq,
                              P, out, algo, (hipStream_t)stream);
}
int combo_msda_backward_f32(const float* grad_out, const float* value, const int64_t* spatial_shapes,
                            const int64_t* level_start_index, const float* sampling_loc, const float* attn_weight,
                            int B, int S, int M, int D, int L, int Lq, int P, float* grad_value,
                            float* grad_sampling_loc, float* grad_attn_weight, int algo, combo_stream_t stream) {
  return msda_backward<float>(grad_out, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, B, S, M,
                              D, L, Lq, P, grad_value, grad_sampling_loc, grad_attn_weight, algo, (hipStream_t)stream);
}
int combo_msda_backward_f64(const double* grad_out, const double* value, const int64_t* spatial_shapes,
                            const int64_t* level_start_index, const double* sampling_loc, const double* attn_weight,
                            int B, int S, int M, int D, int L, int Lq, int P, double* grad_value,
                            double* grad_sampling_loc, double* grad_attn_weight, int algo, combo_stream_t stream) {
  return msda_backward<double>(grad_out, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, B, S, M,
                               D, L, Lq, P, grad_value, grad_sampling_loc, grad_attn_weight, algo,
                               (hipStream_t)stream);
}

}  // extern "C"
