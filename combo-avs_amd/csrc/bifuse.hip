// Token stage of the bilateral audio-visual fusion (reference: fusion_module/utils/fuse_helper.py:155-237, 320-332).
//
// One audio token per frame => both softmaxes run over the HW axis and the projections collapse (see
// combo-avs_amd/modeling/fusion.py).  Per frame b, token i (N = HW tokens, C = 256 channels, 8 heads):
//     xn_i   = LayerNorm(x_i)                         (x already contains +level_embed)
//     s[h,i] = (xn_i + pos_i) . u[b,h,:] + c[b,h]     (clamped to +-5e4 like the reference)
//     p[h,:] = softmax over i
//     y_i    = xn_i + gamma_v * (sum_h pv[h,i] z[b,h,:] + b_ov)          pv = p * dropmask_v
//     pooled[b,h,:] = sum_i pa[h,i] xn_i ;  spa[b,h] = sum_i pa[h,i]      pa = p * dropmask_a
// HBM-bound: forward reads x twice and writes y once (9.6 MB/frame at 56x56x256 fp32) instead of three
// [HW,256]x[256,256] GEMMs + ~20 elementwise launches.
//
// Layout: one WAVE per token (64 lanes x float4 = the 256 channels), 8 waves per workgroup, a workgroup walks a
// contiguous chunk of tokens of ONE frame so that u/z/dpooled (8x256) live in registers (32 floats per lane).
// Cross-lane reductions of the 8 per-head values use a halving butterfly (10 shuffles instead of 48).
// Dropout (train mode, p = 0.1 on both probability tensors, fuse_helper.py:204-205): either injected masks
// (tests) or an in-kernel Philox4x32-10 stream keyed by (seed, frame, token) that the backward regenerates.
#include "combo_common.h"

namespace {

constexpr int C = 256;       // channels (v_dim == embed_dim == 256 in every shipped config)
constexpr int NH = 8;        // heads
constexpr int WAVES = 8;     // waves per workgroup
constexpr int THREADS = WAVES * 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
  return v;
}

// Sum each of 8 per-lane values over the 64 lanes; every lane returns the total of value (lane & 7).
__device__ __forceinline__ float reduce8(const float (&v)[8], int lane) {
  float a[4], b[2], c;
  const bool hi32 = lane & 32, hi16 = lane & 16, hi8 = lane & 8;
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // keep heads {0..3} in the low half, {4..7} in the high half
    const float mine = hi32 ? v[k + 4] : v[k], other = hi32 ? v[k] : v[k + 4];
    a[k] = mine + __shfl_xor(other, 32);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float mine = hi16 ? a[k + 2] : a[k], other = hi16 ? a[k] : a[k + 2];
    b[k] = mine + __shfl_xor(other, 16);
  }
  {
    const float mine = hi8 ? b[1] : b[0], other = hi8 ? b[0] : b[1];
    c = mine + __shfl_xor(other, 8);
  }
  c += __shfl_xor(c, 4);
  c += __shfl_xor(c, 2);
  c += __shfl_xor(c, 1);
  // lane holds head index: bit2 <- hi32, bit1 <- hi16, bit0 <- hi8
  return c;
}
__device__ __forceinline__ int reduce8_head(int lane) { return ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1); }

// broadcast the 8 per-head totals (held as in reduce8) to every lane
__device__ __forceinline__ void bcast8(float c, float (&out)[8]) {
#pragma unroll
  for (int h = 0; h < 8; ++h) out[h] = __shfl(c, ((h >> 2) & 1) * 32 + ((h >> 1) & 1) * 16 + (h & 1) * 8);
}

// ---- Philox4x32-10 -----------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32(unsigned (&ctr)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * ctr[0];
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * ctr[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ ctr[1] ^ k0;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ ctr[3] ^ k1;
    ctr[0] = n0; ctr[1] = (unsigned)p1; ctr[2] = n2; ctr[3] = (unsigned)p0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// dropout multipliers (0 or 1/(1-p)) of the 8 heads of token (b, i); which = 0 (visual side) / 1 (audio side)
__device__ __forceinline__ void drop8(float (&m)[8], const float* inj, long long B, long long N, long long b, long long i,
                                      float p, unsigned long long seed, int which) {
  if (inj) {
#pragma unroll
    for (int h = 0; h < 8; ++h) m[h] = inj[(b * NH + h) * N + i];
    return;
  }
  if (p <= 0.f) {
#pragma unroll
    for (int h = 0; h < 8; ++h) m[h] = 1.f;
    return;
  }
  const float keep = 1.f / (1.f - p);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    unsigned ctr[4] = {(unsigned)i, (unsigned)b, (unsigned)(which * 2 + half), 0x434F4D42u};
    philox4x32(ctr, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
    for (int k = 0; k < 4; ++k) m[half * 4 + k] = ((ctr[k] >> 8) * (1.0f / 16777216.0f) >= p) ? keep : 0.f;
  }
}

// value of `v` in lane k (k uniform over the wave): v_readlane_b32
__device__ __forceinline__ float lane_value(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }

struct TokenLN {
  float4 xn;   // normalised + affine
  float4 xh;   // (x - mean) * rstd
  float rstd;
};

__device__ __forceinline__ TokenLN layer_norm_token(const float4 x, const float4 w, const float4 bb, float eps) {
  const float mean = wave_sum(x.x + x.y + x.z + x.w) * (1.f / C);
  const float4 d = make_float4(x.x - mean, x.y - mean, x.z - mean, x.w - mean);
  const float var = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * (1.f / C);
  TokenLN t;
  t.rstd = rsqrtf(var + eps);
  t.xh = make_float4(d.x * t.rstd, d.y * t.rstd, d.z * t.rstd, d.w * t.rstd);
  t.xn = make_float4(t.xh.x * w.x + bb.x, t.xh.y * w.y + bb.y, t.xh.z * w.z + bb.z, t.xh.w * w.w + bb.w);
  return t;
}

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// Sum the per-wave accumulators acc[h] (h < 8, one float4 per lane) over the 8 waves of the workgroup and store
// [8][C] floats at `dst`.  Two rounds of 4 heads through a 32 KB LDS scratch.
__device__ __forceinline__ void block_reduce_heads(float4 (*red)[4][64], const float4 (&acc)[NH], int wave, int lane,
                                                   float* __restrict__ dst) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) red[wave][k][lane] = acc[half * 4 + k];
    __syncthreads();
    if (threadIdx.x < 256) {  // thread -> (head = tid/64 of this half, lane)
      const int k = threadIdx.x >> 6;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int wv = 0; wv < WAVES; ++wv) {
        const float4 v = red[wv][k][lane];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      reinterpret_cast<float4*>(dst + (half * 4 + k) * C)[lane] = t;
    }
  }
  __syncthreads();
}

// =====================================================================================================
// forward pass 1: scores s[b,h,i] + per-chunk softmax partials (max, sum exp)
// grid = B * chunks; each workgroup handles tokens [chunk*TPC, ...) of frame b
// =====================================================================================================
__global__ void __launch_bounds__(THREADS)
bifuse_scores(const float* __restrict__ x, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
              const float* __restrict__ pos, const float* __restrict__ u, const float* __restrict__ cc, int B, int N,
              int chunks, float* __restrict__ s, float* __restrict__ part /* [B,chunks,NH,2] */, unsigned long long* __restrict__ ts) {
  combo_ts_begin(ts);
  __shared__ float sm[WAVES][NH][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const int tpc = (N + chunks - 1) / chunks;
  const int i0 = chunk * tpc, i1 = min(N, i0 + tpc);
  const float4 w = reinterpret_cast<const float4*>(ln_w)[lane], bb = reinterpret_cast<const float4*>(ln_b)[lane];
  float4 uu[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) uu[h] = reinterpret_cast<const float4*>(u + ((long long)b * NH + h) * C)[lane];
  const int myh = reduce8_head(lane);
  const float myc = cc[b * NH + myh];
  float run_max = -3.0e38f, run_sum = 0.f;
  for (int i = i0 + wave; i < i1; i += WAVES) {
    const float4 xv = reinterpret_cast<const float4*>(x + ((long long)b * N + i) * C)[lane];
    const TokenLN t = layer_norm_token(xv, w, bb, eps);
    const float4 pv = reinterpret_cast<const float4*>(pos + (long long)i * C)[lane];
    const float4 tt = make_float4(t.xn.x + pv.x, t.xn.y + pv.y, t.xn.z + pv.z, t.xn.w + pv.w);
    float d[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) d[h] = dot4(tt, uu[h]);
    float sc = reduce8(d, lane) + myc;
    sc = fminf(fmaxf(sc, -50000.f), 50000.f);  // fuse_helper.py:190-193
    if ((lane & 7) == 0) s[((long long)b * NH + myh) * N + i] = sc;
    const float nm = fmaxf(run_max, sc);
    run_sum = run_sum * __expf(run_max - nm) + __expf(sc - nm);
    run_max = nm;
  }
  if ((lane & 7) == 0) { sm[wave][myh][0] = run_max; sm[wave][myh][1] = run_sum; }
  __syncthreads();
  if (threadIdx.x < NH) {
    float m = -3.0e38f, z = 0.f;
    for (int wv = 0; wv < WAVES; ++wv) {
      const float wm = sm[wv][threadIdx.x][0], wz = sm[wv][threadIdx.x][1];
      const float nm = fmaxf(m, wm);
      z = z * __expf(m - nm) + wz * __expf(wm - nm);
      m = nm;
    }
    part[(((long long)b * chunks + chunk) * NH + threadIdx.x) * 2] = m;
    part[(((long long)b * chunks + chunk) * NH + threadIdx.x) * 2 + 1] = z;
  }
  combo_ts_end(ts);
}

// combine the per-chunk partials -> stat[b,h] = (max, 1/sum)
__global__ void bifuse_softmax_stats(const float* __restrict__ part, int B, int chunks, float* __restrict__ stat) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * NH) return;
  const int b = t / NH, h = t % NH;
  float m = -3.0e38f, z = 0.f;
  for (int k = 0; k < chunks; ++k) {
    const float wm = part[(((long long)b * chunks + k) * NH + h) * 2], wz = part[(((long long)b * chunks + k) * NH + h) * 2 + 1];
    const float nm = fmaxf(m, wm);
    z = z * expf(m - nm) + wz * expf(wm - nm);
    m = nm;
  }
  stat[t * 2] = m;
  stat[t * 2 + 1] = 1.f / z;
}

// =====================================================================================================
// forward pass 2: y, pooled partials
// =====================================================================================================
__global__ void __launch_bounds__(THREADS)
bifuse_apply(const float* __restrict__ x, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
             const float* __restrict__ s, const float* __restrict__ stat, const float* __restrict__ z,
             const float* __restrict__ b_ov, const float* __restrict__ gamma_v, const float* __restrict__ drop_v,
             const float* __restrict__ drop_a, float p_drop, unsigned long long seed, const unsigned long long* seed_step, int B, int N, int chunks,
             float* __restrict__ y, float* __restrict__ pooled_part /* [B,chunks,NH,C] */,
             float* __restrict__ spa_part /* [B,chunks,NH] */, unsigned long long* __restrict__ ts) {
  combo_ts_begin(ts);
  __shared__ float4 red[WAVES][4][64];
  __shared__ float reds[WAVES][NH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (seed_step) seed += *seed_step * 0xD1B54A32D192ED03ull;  // device-side step counter: fresh masks per hipGraph replay
  const int b = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const int tpc = (N + chunks - 1) / chunks;
  const int i0 = chunk * tpc, i1 = min(N, i0 + tpc);
  const float4 w = reinterpret_cast<const float4*>(ln_w)[lane], bb = reinterpret_cast<const float4*>(ln_b)[lane];
  const float4 gv = reinterpret_cast<const float4*>(gamma_v)[lane], bo = reinterpret_cast<const float4*>(b_ov)[lane];
  float4 zz[NH];
  float mx[NH], iz[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    zz[h] = reinterpret_cast<const float4*>(z + ((long long)b * NH + h) * C)[lane];
    mx[h] = stat[(b * NH + h) * 2];
    iz[h] = stat[(b * NH + h) * 2 + 1];
  }
  float4 acc[NH];
  float sacc[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) { acc[h] = make_float4(0.f, 0.f, 0.f, 0.f); sacc[h] = 0.f; }
  // Per-token scalars (8 probabilities, 16 dropout multipliers = 4 Philox blocks, 8 exponentials) are the same in all 64 lanes
  // of the token's wave: computed there they cost ~300 of the ~400 vector instructions per token and made this kernel
  // VALU-bound (180 us for 274 MB).  Instead lane k computes them for the wave's k-th token (up to 64 tokens per round) and
  // the token loop reads them back with v_readlane (100 us).
  for (int base = i0 + wave; base < i1; base += WAVES * 64) {
    const int ik = base + lane * WAVES;  // this lane's token of the round
    float pvl[NH], pal[NH];
    if (ik < i1) {
      float mv[NH], ma[NH];
      drop8(mv, drop_v, B, N, b, ik, p_drop, seed, 0);
      drop8(ma, drop_a, B, N, b, ik, p_drop, seed, 1);
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const float p = __expf(s[((long long)b * NH + h) * N + ik] - mx[h]) * iz[h];
        pvl[h] = p * mv[h];
        pal[h] = p * ma[h];
      }
    } else {
#pragma unroll
      for (int h = 0; h < NH; ++h) pvl[h] = pal[h] = 0.f;
    }
    const int cnt = min(64, (i1 - base + WAVES - 1) / WAVES);
    for (int k = 0; k < cnt; ++k) {
      const int i = base + k * WAVES;
      const float4 xv = reinterpret_cast<const float4*>(x + ((long long)b * N + i) * C)[lane];
      const TokenLN t = layer_norm_token(xv, w, bb, eps);
      float4 o = bo;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const float pv = lane_value(pvl[h], k), pa = lane_value(pal[h], k);
        o.x += pv * zz[h].x; o.y += pv * zz[h].y; o.z += pv * zz[h].z; o.w += pv * zz[h].w;
        acc[h].x += pa * t.xn.x; acc[h].y += pa * t.xn.y; acc[h].z += pa * t.xn.z; acc[h].w += pa * t.xn.w;
        sacc[h] += pa;
      }
      reinterpret_cast<float4*>(y + ((long long)b * N + i) * C)[lane] =
          make_float4(t.xn.x + gv.x * o.x, t.xn.y + gv.y * o.y, t.xn.z + gv.z * o.z, t.xn.w + gv.w * o.w);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int h = 0; h < NH; ++h) reds[wave][h] = sacc[h];
  }
  block_reduce_heads(red, acc, wave, lane, pooled_part + ((long long)b * chunks + chunk) * NH * C);
  if (threadIdx.x < NH) {
    float ss = 0.f;
    for (int wv = 0; wv < WAVES; ++wv) ss += reds[wv][threadIdx.x];
    spa_part[((long long)b * chunks + chunk) * NH + threadIdx.x] = ss;
  }
  combo_ts_end(ts);
}

// =====================================================================================================
// backward pass 1: dp[h,i] (stored), r[h] = sum_i p dp (partials), dz, dgamma_v, db_ov partials
// =====================================================================================================
__global__ void __launch_bounds__(THREADS)
bifuse_bwd1(const float* __restrict__ x, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
            const float* __restrict__ s, const float* __restrict__ stat, const float* __restrict__ z,
            const float* __restrict__ b_ov, const float* __restrict__ gamma_v, const float* __restrict__ drop_v,
            const float* __restrict__ drop_a, float p_drop, unsigned long long seed, const unsigned long long* seed_step, const float* __restrict__ dy,
            const float* __restrict__ dpooled, const float* __restrict__ dspa, int B, int N, int chunks,
            float* __restrict__ dp /* [B,NH,N] */, float* __restrict__ r_part /* [B,chunks,NH] */,
            float* __restrict__ dz_part /* [B,chunks,NH,C] */, float* __restrict__ dgb_part /* [B,chunks,2,C] */, unsigned long long* __restrict__ ts) {
  combo_ts_begin(ts);
  __shared__ float4 red[WAVES][4][64];
  __shared__ float reds[WAVES][NH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (seed_step) seed += *seed_step * 0xD1B54A32D192ED03ull;  // device-side step counter: fresh masks per hipGraph replay
  const int b = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const int tpc = (N + chunks - 1) / chunks;
  const int i0 = chunk * tpc, i1 = min(N, i0 + tpc);
  const float4 w = reinterpret_cast<const float4*>(ln_w)[lane], bb = reinterpret_cast<const float4*>(ln_b)[lane];
  const float4 gv = reinterpret_cast<const float4*>(gamma_v)[lane], bo = reinterpret_cast<const float4*>(b_ov)[lane];
  float4 zz[NH], dpl[NH];
  float mx[NH], iz[NH], dsp[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    zz[h] = reinterpret_cast<const float4*>(z + ((long long)b * NH + h) * C)[lane];
    dpl[h] = reinterpret_cast<const float4*>(dpooled + ((long long)b * NH + h) * C)[lane];
    mx[h] = stat[(b * NH + h) * 2];
    iz[h] = stat[(b * NH + h) * 2 + 1];
    dsp[h] = dspa[b * NH + h];
  }
  const int myh = reduce8_head(lane);
  float4 dzacc[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) dzacc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 dgacc = make_float4(0.f, 0.f, 0.f, 0.f), dbacc = make_float4(0.f, 0.f, 0.f, 0.f);
  float racc = 0.f;  // for head myh (valid on lanes with (lane&7)==0)
  // (per-token scalars computed by lane k for the wave's k-th token and read back with v_readlane: see bifuse_apply)
  for (int base = i0 + wave; base < i1; base += WAVES * 64) {
    const int ik = base + lane * WAVES;
    float mvl[NH], mal[NH], prl[NH];
    if (ik < i1) {
      drop8(mvl, drop_v, B, N, b, ik, p_drop, seed, 0);
      drop8(mal, drop_a, B, N, b, ik, p_drop, seed, 1);
#pragma unroll
      for (int h = 0; h < NH; ++h) prl[h] = __expf(s[((long long)b * NH + h) * N + ik] - mx[h]) * iz[h];
    } else {
#pragma unroll
      for (int h = 0; h < NH; ++h) mvl[h] = mal[h] = prl[h] = 0.f;
    }
    const int cnt = min(64, (i1 - base + WAVES - 1) / WAVES);
    for (int k = 0; k < cnt; ++k) {
      const int i = base + k * WAVES;
      const float4 xv = reinterpret_cast<const float4*>(x + ((long long)b * N + i) * C)[lane];
      const TokenLN t = layer_norm_token(xv, w, bb, eps);
      const float4 dyv = reinterpret_cast<const float4*>(dy + ((long long)b * N + i) * C)[lane];
      const float4 g = make_float4(dyv.x * gv.x, dyv.y * gv.y, dyv.z * gv.z, dyv.w * gv.w);
      float mv[NH], ma[NH], pr[NH], d1[NH], d2[NH];
      float4 o = bo;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        mv[h] = lane_value(mvl[h], k); ma[h] = lane_value(mal[h], k); pr[h] = lane_value(prl[h], k);
        const float pv = pr[h] * mv[h];
        o.x += pv * zz[h].x; o.y += pv * zz[h].y; o.z += pv * zz[h].z; o.w += pv * zz[h].w;
        dzacc[h].x += pv * g.x; dzacc[h].y += pv * g.y; dzacc[h].z += pv * g.z; dzacc[h].w += pv * g.w;
        d1[h] = dot4(g, zz[h]);        // d pv[h]
        d2[h] = dot4(t.xn, dpl[h]);    // d pa[h] (without dspa)
      }
      dgacc.x += dyv.x * o.x; dgacc.y += dyv.y * o.y; dgacc.z += dyv.z * o.z; dgacc.w += dyv.w * o.w;
      dbacc.x += g.x; dbacc.y += g.y; dbacc.z += g.z; dbacc.w += g.w;
      const float r1 = reduce8(d1, lane), r2 = reduce8(d2, lane);
      float mvh = mv[0], mah = ma[0], prh = pr[0], dsh = dsp[0];
#pragma unroll
      for (int h = 1; h < NH; ++h)
        if (myh == h) { mvh = mv[h]; mah = ma[h]; prh = pr[h]; dsh = dsp[h]; }
      const float dph = r1 * mvh + (r2 + dsh) * mah;
      if ((lane & 7) == 0) {
        dp[((long long)b * NH + myh) * N + i] = dph;
        racc += prh * dph;
      }
    }
  }
  // ---- workgroup reductions -> partial buffers ----
  if ((lane & 7) == 0) reds[wave][myh] = racc;
  block_reduce_heads(red, dzacc, wave, lane, dz_part + ((long long)b * chunks + chunk) * NH * C);
  if (threadIdx.x < NH) {
    float ss = 0.f;
    for (int wv = 0; wv < WAVES; ++wv) ss += reds[wv][threadIdx.x];
    r_part[((long long)b * chunks + chunk) * NH + threadIdx.x] = ss;
  }
  __syncthreads();
  red[wave][0][lane] = dgacc;
  red[wave][1][lane] = dbacc;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int k = threadIdx.x >> 6;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int wv = 0; wv < WAVES; ++wv) {
      const float4 v = red[wv][k][lane];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    reinterpret_cast<float4*>(dgb_part + (((long long)b * chunks + chunk) * 2 + k) * C)[lane] = t;
  }
  combo_ts_end(ts);
}

// =====================================================================================================
// backward pass 2: ds, dx (through LayerNorm), du, dc, d ln_w, d ln_b partials
// =====================================================================================================
__global__ void __launch_bounds__(THREADS)
bifuse_bwd2(const float* __restrict__ x, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
            const float* __restrict__ pos, const float* __restrict__ s, const float* __restrict__ stat,
            const float* __restrict__ u, const float* __restrict__ drop_a, float p_drop, unsigned long long seed, const unsigned long long* seed_step,
            const float* __restrict__ dy, const float* __restrict__ dpooled, const float* __restrict__ dp,
            const float* __restrict__ rtot /* [B,NH] */, int B, int N, int chunks, float* __restrict__ dx,
            float* __restrict__ du_part /* [B,chunks,NH,C] */, float* __restrict__ dc_part /* [B,chunks,NH] */,
            float* __restrict__ dln_part /* [B,chunks,2,C] */, unsigned long long* __restrict__ ts) {
  combo_ts_begin(ts);
  __shared__ float4 red[WAVES][4][64];
  __shared__ float reds[WAVES][NH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (seed_step) seed += *seed_step * 0xD1B54A32D192ED03ull;  // device-side step counter: fresh masks per hipGraph replay
  const int b = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const int tpc = (N + chunks - 1) / chunks;
  const int i0 = chunk * tpc, i1 = min(N, i0 + tpc);
  const float4 w = reinterpret_cast<const float4*>(ln_w)[lane], bb = reinterpret_cast<const float4*>(ln_b)[lane];
  float4 uu[NH], dpl[NH];
  float mx[NH], iz[NH], rt[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    uu[h] = reinterpret_cast<const float4*>(u + ((long long)b * NH + h) * C)[lane];
    dpl[h] = reinterpret_cast<const float4*>(dpooled + ((long long)b * NH + h) * C)[lane];
    mx[h] = stat[(b * NH + h) * 2];
    iz[h] = stat[(b * NH + h) * 2 + 1];
    rt[h] = rtot[b * NH + h];
  }
  float4 duacc[NH];
  float dcacc[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) { duacc[h] = make_float4(0.f, 0.f, 0.f, 0.f); dcacc[h] = 0.f; }
  float4 dwacc = make_float4(0.f, 0.f, 0.f, 0.f), dbacc = make_float4(0.f, 0.f, 0.f, 0.f);
  // (per-token scalars - the audio-side probability pa and the score gradient ds of the 8 heads - computed by lane k for the
  // wave's k-th token and read back with v_readlane: see bifuse_apply)
  for (int base = i0 + wave; base < i1; base += WAVES * 64) {
    const int ik = base + lane * WAVES;
    float pal[NH], dsl[NH];
    if (ik < i1) {
      float ma[NH];
      drop8(ma, drop_a, B, N, b, ik, p_drop, seed, 1);
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const float sc = s[((long long)b * NH + h) * N + ik];
        const float p = __expf(sc - mx[h]) * iz[h];
        // the clamp of the scores has zero gradient outside +-5e4
        dsl[h] = (sc > -50000.f && sc < 50000.f) ? p * (dp[((long long)b * NH + h) * N + ik] - rt[h]) : 0.f;
        pal[h] = p * ma[h];
      }
    } else {
#pragma unroll
      for (int h = 0; h < NH; ++h) pal[h] = dsl[h] = 0.f;
    }
    const int cnt = min(64, (i1 - base + WAVES - 1) / WAVES);
    for (int k = 0; k < cnt; ++k) {
      const int i = base + k * WAVES;
      const float4 xv = reinterpret_cast<const float4*>(x + ((long long)b * N + i) * C)[lane];
      const TokenLN t = layer_norm_token(xv, w, bb, eps);
      const float4 pv = reinterpret_cast<const float4*>(pos + (long long)i * C)[lane];
      const float4 tt = make_float4(t.xn.x + pv.x, t.xn.y + pv.y, t.xn.z + pv.z, t.xn.w + pv.w);
      float4 dxn = reinterpret_cast<const float4*>(dy + ((long long)b * N + i) * C)[lane];
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const float ds = lane_value(dsl[h], k), pa = lane_value(pal[h], k);
        dxn.x += pa * dpl[h].x + ds * uu[h].x; dxn.y += pa * dpl[h].y + ds * uu[h].y;
        dxn.z += pa * dpl[h].z + ds * uu[h].z; dxn.w += pa * dpl[h].w + ds * uu[h].w;
        duacc[h].x += ds * tt.x; duacc[h].y += ds * tt.y; duacc[h].z += ds * tt.z; duacc[h].w += ds * tt.w;
        dcacc[h] += ds;
      }
      // LayerNorm backward: dx = rstd * (gw - mean(gw) - xh * mean(gw * xh)),  gw = dxn * w
      dwacc.x += dxn.x * t.xh.x; dwacc.y += dxn.y * t.xh.y; dwacc.z += dxn.z * t.xh.z; dwacc.w += dxn.w * t.xh.w;
      dbacc.x += dxn.x; dbacc.y += dxn.y; dbacc.z += dxn.z; dbacc.w += dxn.w;
      const float4 gw = make_float4(dxn.x * w.x, dxn.y * w.y, dxn.z * w.z, dxn.w * w.w);
      const float m1 = wave_sum(gw.x + gw.y + gw.z + gw.w) * (1.f / C);
      const float m2 = wave_sum(gw.x * t.xh.x + gw.y * t.xh.y + gw.z * t.xh.z + gw.w * t.xh.w) * (1.f / C);
      reinterpret_cast<float4*>(dx + ((long long)b * N + i) * C)[lane] =
          make_float4(t.rstd * (gw.x - m1 - t.xh.x * m2), t.rstd * (gw.y - m1 - t.xh.y * m2),
                      t.rstd * (gw.z - m1 - t.xh.z * m2), t.rstd * (gw.w - m1 - t.xh.w * m2));
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int h = 0; h < NH; ++h) reds[wave][h] = dcacc[h];
  }
  block_reduce_heads(red, duacc, wave, lane, du_part + ((long long)b * chunks + chunk) * NH * C);
  if (threadIdx.x < NH) {
    float ss = 0.f;
    for (int wv = 0; wv < WAVES; ++wv) ss += reds[wv][threadIdx.x];
    dc_part[((long long)b * chunks + chunk) * NH + threadIdx.x] = ss;
  }
  __syncthreads();
  red[wave][0][lane] = dwacc;
  red[wave][1][lane] = dbacc;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int k = threadIdx.x >> 6;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int wv = 0; wv < WAVES; ++wv) {
      const float4 v = red[wv][k][lane];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    reinterpret_cast<float4*>(dln_part + (((long long)b * chunks + chunk) * 2 + k) * C)[lane] = t;
  }
  combo_ts_end(ts);
}

}  // namespace

extern "C" {

int combo_bifuse_chunks(int B, int N) {
  // enough workgroups to fill 256 CUs ~4x, at least 64 tokens (8 per wave) per workgroup
  int chunks = (1024 + B - 1) / B;
  const int maxc = (N + 63) / 64;
  if (chunks > maxc) chunks = maxc;
  if (chunks < 1) chunks = 1;
  return chunks;
}

int combo_bifuse_forward_f32(const float* x, const float* ln_w, const float* ln_b, float eps, const float* pos,
                             const float* u, const float* c, const float* z, const float* b_ov, const float* gamma_v,
                             const float* drop_v, const float* drop_a, float p_drop, unsigned long long seed, const unsigned long long* seed_step, int B,
                             int N, int Cch, int heads, float* y, float* scores, float* stat, float* part_ws,
                             float* pooled_part, float* spa_part, combo_stream_t stream) {
  if (!x || !ln_w || !ln_b || !pos || !u || !c || !z || !b_ov || !gamma_v || !y || !scores || !stat || !part_ws ||
      !pooled_part || !spa_part || B <= 0 || N <= 0 || Cch != C || heads != NH)
    return COMBO_EINVAL;
  const int chunks = combo_bifuse_chunks(B, N);
  hipStream_t st = (hipStream_t)stream;
  // device-side timing slots (bench.py's `other_kernels.bifuse`); work = algorithmic HBM bytes of the launch
  const double act = (double)B * N * C * 4.0, sc = (double)B * NH * N * 4.0, ps = (double)N * C * 4.0;
  hipLaunchKernelGGL(bifuse_scores, dim3(B * chunks), dim3(THREADS), 0, st, x, ln_w, ln_b, eps, pos, u, c, B, N, chunks,
                     scores, part_ws, combo_timing_next_slot(COMBO_TS_BIFUSE, act + ps + sc, act + ps + sc));
  hipLaunchKernelGGL(bifuse_softmax_stats, dim3((B * NH + 63) / 64), dim3(64), 0, st, part_ws, B, chunks, stat);
  hipLaunchKernelGGL(bifuse_apply, dim3(B * chunks), dim3(THREADS), 0, st, x, ln_w, ln_b, eps, scores, stat, z, b_ov,
                     gamma_v, drop_v, drop_a, p_drop, seed, seed_step, B, N, chunks, y, pooled_part, spa_part,
                     combo_timing_next_slot(COMBO_TS_BIFUSE, 2.0 * act + sc, 2.0 * act + sc));
  return (int)hipGetLastError();
}

int combo_bifuse_backward1_f32(const float* x, const float* ln_w, const float* ln_b, float eps, const float* scores,
                               const float* stat, const float* z, const float* b_ov, const float* gamma_v,
                               const float* drop_v, const float* drop_a, float p_drop, unsigned long long seed, const unsigned long long* seed_step,
                               const float* dy, const float* dpooled, const float* dspa, int B, int N, int Cch,
                               int heads, float* dp, float* r_part, float* dz_part, float* dgb_part,
                               combo_stream_t stream) {
  if (!x || !scores || !stat || !z || !dy || !dpooled || !dspa || !dp || !r_part || !dz_part || !dgb_part || B <= 0 ||
      N <= 0 || Cch != C || heads != NH)
    return COMBO_EINVAL;
  const int chunks = combo_bifuse_chunks(B, N);
  hipLaunchKernelGGL(bifuse_bwd1, dim3(B * chunks), dim3(THREADS), 0, (hipStream_t)stream, x, ln_w, ln_b, eps, scores,
                     stat, z, b_ov, gamma_v, drop_v, drop_a, p_drop, seed, seed_step, dy, dpooled, dspa, B, N, chunks, dp, r_part,
                     dz_part, dgb_part,
                     combo_timing_next_slot(COMBO_TS_BIFUSE, 2.0 * B * N * C * 4.0 + 2.0 * B * NH * N * 4.0,
                                            2.0 * B * N * C * 4.0 + 2.0 * B * NH * N * 4.0));
  return (int)hipGetLastError();
}

int combo_bifuse_backward2_f32(const float* x, const float* ln_w, const float* ln_b, float eps, const float* pos,
                               const float* scores, const float* stat, const float* u, const float* drop_a,
                               float p_drop, unsigned long long seed, const unsigned long long* seed_step, const float* dy, const float* dpooled,
                               const float* dp, const float* rtot, int B, int N, int Cch, int heads, float* dx,
                               float* du_part, float* dc_part, float* dln_part, combo_stream_t stream) {
  if (!x || !pos || !scores || !stat || !u || !dy || !dpooled || !dp || !rtot || !dx || !du_part || !dc_part ||
      !dln_part || B <= 0 || N <= 0 || Cch != C || heads != NH)
    return COMBO_EINVAL;
  const int chunks = combo_bifuse_chunks(B, N);
  hipLaunchKernelGGL(bifuse_bwd2, dim3(B * chunks), dim3(THREADS), 0, (hipStream_t)stream, x, ln_w, ln_b, eps, pos,
                     scores, stat, u, drop_a, p_drop, seed, seed_step, dy, dpooled, dp, rtot, B, N, chunks, dx, du_part, dc_part,
                     dln_part,
                     combo_timing_next_slot(COMBO_TS_BIFUSE, 3.0 * B * N * C * 4.0 + (double)N * C * 4.0 + 2.0 * B * NH * N * 4.0,
                                            3.0 * B * N * C * 4.0 + (double)N * C * 4.0 + 2.0 * B * NH * N * 4.0));
  return (int)hipGetLastError();
}

}  // extern "C"
