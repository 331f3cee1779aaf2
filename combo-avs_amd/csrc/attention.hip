// Masked multi-head attention of the transformer decoder (reference: nn.MultiheadAttention as called at
// transformer_decoder/transformer_decoder.py:99-118 (cross-attention, bool mask [BT*8, 100, hw] shared by the 8 heads) and
// :50-58 (self-attention)), forward and backward, head_dim = 32.
//
// Shapes of this task: 100 queries x {49, 196, 784} keys (cross) or 100 keys (self) x 32 channels, BT*8 = 320 (frame, head)
// pairs per layer - tiny GEMMs that a library runs as generic flash attention (aotriton: 156 us forward / 351 us backward
// for the 784-key layers).  Here everything is shaped around the fp32 matrix instruction v_mfma_f32_32x32x2_f32 (exact
// fp32: the attention output feeds the mask logits that are thresholded at 0, see gemm_f32.hip):
//   * the score tile is computed TRANSPOSED, S^T[32 keys x 32 queries] = K_tile . Q^T: in the D layout a lane then owns ONE
//     query column (lane & 31) and 16 of the 32 keys, so the online-softmax statistics are per-lane scalars plus one exchange
//     with lane ^ 32 - no 32-lane row reductions;
//   * P^T in D layout IS the B operand of O^T[32 d x 32 q] += V^T . P^T when the contraction index is enumerated as
//     k-slot (step e, half g) <-> key (e & 3) + 8 (e >> 2) + 4 g; the A operand V^T is loaded with the same enumeration
//     (lane (d, g) reads V[key(e, g)][d]: two coalesced 128-byte rows per load).  No transpose, no LDS;
//   * one wave = 32 queries of one (frame, head) pair, operands straight from L2 (K / V of a pair are 100 KB and are read by
//     its 4 query waves), next tile prefetched into registers during the MFMAs of the current one;
//   * the mask is ONE byte per (frame, query, key), shared by the heads (the reference materialises 8 copies), rows padded to
//     a multiple of 4 bytes so a lane reads the 4 keys of a register group with one dword load.
// Backward, two passes that each recompute the scores in the orientation they need (no atomics, no dQ round trip):
//   pass A (a wave owns 32 queries, walks the key tiles): S^T, dP^T = V . dO^T, dS^T, dQ^T += K^T . dS^T;
//   pass B (a wave owns 32 keys, walks the query tiles):   S, dP = dO . V^T, dS, dV^T += dO^T . P, dK^T += Q^T . dS.
#include <math.h>

#include "combo_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int kD = 32;          // head_dim
constexpr float kNeg = -1e30f;  // "minus infinity" that stays finite under subtraction
__device__ __attribute__((aligned(16))) unsigned char g_no_mask[16];  // zero-initialised: the "mask" of unmasked attention (pitch 4)

__device__ __forceinline__ float xor32(float v) {  // value of lane ^ 32
  return __shfl_xor(v, 32, 64);
}
// key / query index inside a 32-row tile held by register e of lane-half g in the MFMA D layout
__device__ __forceinline__ constexpr int row_of(int e, int g) { return (e & 3) + 8 * (e >> 2) + 4 * g; }

struct AttnArgs {
  const float* q; const float* k; const float* v;  // [B, L, ld] rows, head h at column h * 32
  long long ldq, ldk, ldv;
  const unsigned char* mask;  // [B, Lq, pitch] bytes (1 = blocked) or nullptr
  int pitch;
  int B, H, Lq, Lk;
  float scale;
  float* out;  // forward: [B, Lq, H*32]
  float* lse;  // [B, H, Lq]  log-sum-exp of the scaled, masked scores
  // backward
  const float* dout;   // [B, Lq, H*32]
  const float* delta;  // [B, H, Lq]  sum_d dO * O
  float* dq; float* dk; float* dv;  // [B, Lq, H*32], [B, Lk, H*32], [B, Lk, H*32]
  unsigned long long* ts;
};

// 16 consecutive channels (half g) of one row as 4 float4
__device__ __forceinline__ void load_row16(const float* row, int g, float (&f)[16], float mul = 1.f) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f4v t = *reinterpret_cast<const f4v*>(row + 16 * g + 4 * j);
    f[4 * j] = t.x * mul; f[4 * j + 1] = t.y * mul; f[4 * j + 2] = t.z * mul; f[4 * j + 3] = t.w * mul;
  }
}

// blocked flags of the 16 (query, key) cells a lane holds: register group j = e >> 2 covers 4 consecutive columns
__device__ __forceinline__ void load_mask_t(const unsigned char* mrow, int pitch, int col0, int g, unsigned (&mw)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int off = min(col0 + 8 * j + 4 * g, pitch - 4);
    mw[j] = *reinterpret_cast<const unsigned*>(mrow + off);
  }
}

// ------------------------------------------------------------------------------------------------ forward
__global__ void __launch_bounds__(256, 2)
attn_fwd_kernel(const AttnArgs a) {
  combo_ts_begin(a.ts);
  const int pair = blockIdx.x, b = pair / a.H, h = pair - b * a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, g = lane >> 5;  // column (query) / row-of-A (key or d) index and k-half
  const int n_qt = (a.Lq + 31) / 32, n_kt = (a.Lk + 31) / 32;
  for (int qt = wave; qt < n_qt; qt += 4) {
    const int qi = qt * 32 + c;
    const bool q_ok = qi < a.Lq;
    const int qc = min(qi, a.Lq - 1);
    float qf[16];
    load_row16(a.q + ((long long)b * a.Lq + qc) * a.ldq + h * kD, g, qf, a.scale);
    const unsigned char* mrow = a.mask ? a.mask + ((long long)b * a.Lq + qc) * a.pitch : g_no_mask;
    const int pitch = a.mask ? a.pitch : 4;
    const float* kb = a.k + (long long)b * a.Lk * a.ldk + h * kD;
    const float* vb = a.v + (long long)b * a.Lk * a.ldv + h * kD;
    f32x16 o;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.f;
    float m = kNeg, l = 0.f;
    // operands of the tile in flight
    float kf[16], vf[16];
    unsigned mw[4];
    auto load_tile = [&](int kt) __attribute__((always_inline)) {
      load_row16(kb + (long long)min(kt * 32 + c, a.Lk - 1) * a.ldk, g, kf);
#pragma unroll
      for (int e = 0; e < 16; ++e) vf[e] = vb[(long long)min(kt * 32 + row_of(e, g), a.Lk - 1) * a.ldv + c];
      load_mask_t(mrow, pitch, kt * 32, g, mw);
    };
    load_tile(0);
    for (int kt = 0; kt < n_kt; ++kt) {
      // S^T tile: rows = keys, columns = queries
      f32x16 s;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t], qf[t], s, 0, 0, 0);
      float vcur[16];
      unsigned mcur[4];
#pragma unroll
      for (int e = 0; e < 16; ++e) vcur[e] = vf[e];
#pragma unroll
      for (int j = 0; j < 4; ++j) mcur[j] = mw[j];
      if (kt + 1 < n_kt) load_tile(kt + 1);  // in flight during the softmax and the P.V MFMAs below
      // branch-free masking: blocked cells (mask byte = 1, or a key beyond Lk in the last tile) get the score kNeg
      float mx = kNeg;
      const int kbase = kt * 32 + 4 * g;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const unsigned bit = ((mcur[e >> 2] >> (8 * (e & 3))) & 1u) | (unsigned)(kbase + (e & 3) + 8 * (e >> 2) >= a.Lk);
        s[e] = bit ? kNeg : s[e];
        mx = fmaxf(mx, s[e]);
      }
      mx = fmaxf(mx, xor32(mx));
      const float m_new = fmaxf(m, mx);
      const float alpha = __expf(m - m_new);
      m = m_new;
      float ps = 0.f;
      float p[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float ex = __expf(s[e] - m_new);  // a blocked cell: exp(kNeg - m_new) = 0, or 1 while m_new is still kNeg
        p[e] = s[e] > 0.5f * kNeg ? ex : 0.f;
        ps += p[e];
        o[e] *= alpha;
      }
      l = l * alpha + ps;
#pragma unroll
      for (int e = 0; e < 16; ++e) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vcur[e], p[e], o, 0, 0, 0);
    }
    const float lt = l + xor32(l);
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    if (q_ok) {
      float* orow = a.out + ((long long)b * a.Lq + qi) * (a.H * kD) + h * kD;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f4v*>(orow + 8 * j + 4 * g) = f4v{o[4 * j] * inv, o[4 * j + 1] * inv, o[4 * j + 2] * inv, o[4 * j + 3] * inv};
      if (g == 0 && a.lse) a.lse[((long long)b * a.H + h) * a.Lq + qi] = m + __logf(lt);
    }
  }
  combo_ts_end(a.ts);
}

// delta[b, h, q] = sum_d dO[b, q, h*32 + d] * O[b, q, h*32 + d]
__global__ void __launch_bounds__(256)
attn_delta_kernel(const float* __restrict__ dout, const float* __restrict__ out, int B, int H, int Lq, float* __restrict__ delta) {
  const long long t = blockIdx.x * 256LL + threadIdx.x;  // one thread per (b, q, h)
  if (t >= (long long)B * Lq * H) return;
  const int h = (int)(t % H);
  const long long bq = t / H;
  const float* a = dout + bq * (H * kD) + h * kD;
  const float* o = out + bq * (H * kD) + h * kD;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const f4v x = *reinterpret_cast<const f4v*>(a + 4 * j), y = *reinterpret_cast<const f4v*>(o + 4 * j);
    s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
  }
  const int b = (int)(bq / Lq), q = (int)(bq - (long long)b * Lq);
  delta[((long long)b * H + h) * Lq + q] = s;
}

// ------------------------------------------------------------------------------------------------ backward, pass A: dQ
__global__ void __launch_bounds__(256, 2)
attn_bwd_dq_kernel(const AttnArgs a) {
  combo_ts_begin(a.ts);
  const int pair = blockIdx.x, b = pair / a.H, h = pair - b * a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, g = lane >> 5;
  const int n_qt = (a.Lq + 31) / 32, n_kt = (a.Lk + 31) / 32;
  for (int qt = wave; qt < n_qt; qt += 4) {
    const int qi = qt * 32 + c;
    const bool q_ok = qi < a.Lq;
    const int qc = min(qi, a.Lq - 1);
    float qf[16], dof[16];
    load_row16(a.q + ((long long)b * a.Lq + qc) * a.ldq + h * kD, g, qf, a.scale);
    load_row16(a.dout + ((long long)b * a.Lq + qc) * (a.H * kD) + h * kD, g, dof);
    const float lse = a.lse[((long long)b * a.H + h) * a.Lq + qc];
    const float dl = a.delta[((long long)b * a.H + h) * a.Lq + qc];
    const unsigned char* mrow = a.mask ? a.mask + ((long long)b * a.Lq + qc) * a.pitch : g_no_mask;
    const int pitch = a.mask ? a.pitch : 4;
    const float* kb = a.k + (long long)b * a.Lk * a.ldk + h * kD;
    const float* vb = a.v + (long long)b * a.Lk * a.ldv + h * kD;
    f32x16 dq;
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[e] = 0.f;
    float kf[16], vr[16], ktf[16];
    unsigned mw[4];
    auto load_tile = [&](int kt) __attribute__((always_inline)) {
      const int kr = min(kt * 32 + c, a.Lk - 1);
      load_row16(kb + (long long)kr * a.ldk, g, kf);
      load_row16(vb + (long long)kr * a.ldv, g, vr);
#pragma unroll
      for (int e = 0; e < 16; ++e) ktf[e] = kb[(long long)min(kt * 32 + row_of(e, g), a.Lk - 1) * a.ldk + c];
      load_mask_t(mrow, pitch, kt * 32, g, mw);
    };
    load_tile(0);
    for (int kt = 0; kt < n_kt; ++kt) {
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
      for (int t = 0; t < 16; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t], qf[t], s, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 16; ++t) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[t], dof[t], dp, 0, 0, 0);
      float kcur[16];
      unsigned mcur[4];
#pragma unroll
      for (int e = 0; e < 16; ++e) kcur[e] = ktf[e];
#pragma unroll
      for (int j = 0; j < 4; ++j) mcur[j] = mw[j];
      if (kt + 1 < n_kt) load_tile(kt + 1);  // in flight during the dS arithmetic and the dQ MFMAs below
      float ds[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const unsigned bit = ((mcur[e >> 2] >> (8 * (e & 3))) & 1u) | (unsigned)(kt * 32 + row_of(e, g) >= a.Lk) | (unsigned)(!q_ok);
        const float p = bit ? 0.f : __expf(s[e] - lse);
        ds[e] = p * (dp[e] - dl);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) dq = __builtin_amdgcn_mfma_f32_32x32x2f32(kcur[e], ds[e], dq, 0, 0, 0);
    }
    if (q_ok) {
      float* row = a.dq + ((long long)b * a.Lq + qi) * (a.H * kD) + h * kD;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f4v*>(row + 8 * j + 4 * g) =
            f4v{dq[4 * j] * a.scale, dq[4 * j + 1] * a.scale, dq[4 * j + 2] * a.scale, dq[4 * j + 3] * a.scale};
    }
  }
  combo_ts_end(a.ts);
}

// ------------------------------------------------------------------------------------------------ backward, pass B: dK, dV
__global__ void __launch_bounds__(256, 1)
attn_bwd_dkv_kernel(const AttnArgs a) {
  combo_ts_begin(a.ts);
  const int pair = blockIdx.x, b = pair / a.H, h = pair - b * a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, g = lane >> 5;  // here the column index is the KEY
  const int n_qt = (a.Lq + 31) / 32, n_kt = (a.Lk + 31) / 32;
  const int kt = blockIdx.y * 4 + wave;
  if (kt < n_kt) {
    const int ki = kt * 32 + c;
    const bool k_ok = ki < a.Lk;
    const int kc = min(ki, a.Lk - 1);
    const float* kb = a.k + (long long)b * a.Lk * a.ldk + h * kD;
    const float* vb = a.v + (long long)b * a.Lk * a.ldv + h * kD;
    float kcol[16], vcol[16];
    load_row16(kb + (long long)kc * a.ldk, g, kcol);
    load_row16(vb + (long long)kc * a.ldv, g, vcol);
    const float* qb = a.q + (long long)b * a.Lq * a.ldq + h * kD;
    const float* dob = a.dout + (long long)b * a.Lq * (a.H * kD) + h * kD;
    const float* lseb = a.lse + ((long long)b * a.H + h) * a.Lq;
    const float* dlb = a.delta + ((long long)b * a.H + h) * a.Lq;
    const unsigned char* mbase = a.mask ? a.mask + (long long)b * a.Lq * a.pitch : g_no_mask;
    const long long mpitch = a.mask ? a.pitch : 0;
    const int mcol = a.mask ? kc : 0;
    f32x16 dk, dv;
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
    // lse / delta rows are padded reads: the buffers hold Lq floats per (b, h); indices are clamped per group of 4
    float qrow[16], dorow[16], qtf[16], dotf[16], lse[16], dl[16];
    unsigned char mb[16];
    auto load_qtile = [&](int qt) __attribute__((always_inline)) {
      const int qr = min(qt * 32 + c, a.Lq - 1);  // this lane's row of the A operands (a query)
      load_row16(qb + (long long)qr * a.ldq, g, qrow, a.scale);
      load_row16(dob + (long long)qr * (a.H * kD), g, dorow);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int q = min(qt * 32 + row_of(e, g), a.Lq - 1);
        qtf[e] = qb[(long long)q * a.ldq + c] * a.scale;
        dotf[e] = dob[(long long)q * (a.H * kD) + c];
        lse[e] = lseb[q];
        dl[e] = dlb[q];
        mb[e] = mbase[(long long)q * mpitch + mcol];
      }
    };
    load_qtile(0);
    for (int qt = 0; qt < n_qt; ++qt) {
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
      for (int t = 0; t < 16; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[t], kcol[t], s, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 16; ++t) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dorow[t], vcol[t], dp, 0, 0, 0);
      float p[16], ds[16], qtc[16], dotc[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const unsigned bit = (unsigned)(mb[e] & 1) | (unsigned)(qt * 32 + row_of(e, g) >= a.Lq) | (unsigned)(!k_ok);
        p[e] = bit ? 0.f : __expf(s[e] - lse[e]);
        ds[e] = p[e] * (dp[e] - dl[e]);
        qtc[e] = qtf[e];
        dotc[e] = dotf[e];
      }
      if (qt + 1 < n_qt) load_qtile(qt + 1);  // in flight during the dV / dK MFMAs below
#pragma unroll
      for (int e = 0; e < 16; ++e) dv = __builtin_amdgcn_mfma_f32_32x32x2f32(dotc[e], p[e], dv, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 16; ++e) dk = __builtin_amdgcn_mfma_f32_32x32x2f32(qtc[e], ds[e], dk, 0, 0, 0);
    }
    if (k_ok) {
      float* rk = a.dk + ((long long)b * a.Lk + ki) * (a.H * kD) + h * kD;
      float* rv = a.dv + ((long long)b * a.Lk + ki) * (a.H * kD) + h * kD;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<f4v*>(rk + 8 * j + 4 * g) = f4v{dk[4 * j], dk[4 * j + 1], dk[4 * j + 2], dk[4 * j + 3]};
        *reinterpret_cast<f4v*>(rv + 8 * j + 4 * g) = f4v{dv[4 * j], dv[4 * j + 1], dv[4 * j + 2], dv[4 * j + 3]};
      }
    }
  }
  combo_ts_end(a.ts);
}

bool common_ok(const float* q, long long ldq, const float* k, long long ldk, const float* v, long long ldv, int B, int H, int Lq, int Lk,
               const unsigned char* mask, int pitch) {
  return q && k && v && B > 0 && H > 0 && Lq > 0 && Lk > 0 && ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 &&
         !((uintptr_t)q & 15) && !((uintptr_t)k & 15) && !((uintptr_t)v & 15) &&
         (!mask || (pitch % 4 == 0 && pitch >= Lk && pitch >= 4 && !((uintptr_t)mask & 3)));
}

}  // namespace

extern "C" int combo_attention_forward_f32(const float* q, long long ldq, const float* k, long long ldk, const float* v, long long ldv,
                                           const unsigned char* blocked, int pitch, int B, int H, int Lq, int Lk, float scale,
                                           float* out, float* lse, combo_stream_t stream) {
  if (!common_ok(q, ldq, k, ldk, v, ldv, B, H, Lq, Lk, blocked, pitch) || !out || ((uintptr_t)out & 15)) return COMBO_EINVAL;
  AttnArgs a{q, k, v, ldq, ldk, ldv, blocked, pitch, B, H, Lq, Lk, scale, out, lse, nullptr, nullptr, nullptr, nullptr, nullptr,
             combo_timing_next_slot(COMBO_TS_ATTN_FWD, 4.0 * B * H * (double)Lq * Lk * kD)};
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" int combo_attention_backward_f32(const float* q, long long ldq, const float* k, long long ldk, const float* v, long long ldv,
                                            const unsigned char* blocked, int pitch, int B, int H, int Lq, int Lk, float scale,
                                            const float* out, const float* lse, const float* dout, float* delta_ws, float* dq,
                                            float* dk, float* dv, combo_stream_t stream) {
  if (!common_ok(q, ldq, k, ldk, v, ldv, B, H, Lq, Lk, blocked, pitch) || !out || !lse || !dout || !delta_ws || !dq || !dk || !dv ||
      (((uintptr_t)out | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15))
    return COMBO_EINVAL;
  const long long n = (long long)B * Lq * H;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, out, B, H, Lq, delta_ws);
  AttnArgs a{q, k, v, ldq, ldk, ldv, blocked, pitch, B, H, Lq, Lk, scale, nullptr, const_cast<float*>(lse), dout, delta_ws, dq, dk, dv,
             combo_timing_next_slot(COMBO_TS_ATTN_BWD, 6.0 * B * H * (double)Lq * Lk * kD)};
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, a);
  a.ts = combo_timing_next_slot(COMBO_TS_ATTN_BWD, 8.0 * B * H * (double)Lq * Lk * kD);
  const int n_kt = (Lk + 31) / 32;
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(B * H, (n_kt + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
