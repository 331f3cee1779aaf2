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
//     k-slot (step e, half g) <-> key (e & 3) + 8 (e >> 2) + 4 g; the A operand V^T is read with the same enumeration
//     (lane (d, g) reads V[key(e, g)][d]) - no transpose;
//   * operand tiles (32 rows x 32 channels) are staged by LDS-DMA with the 16-byte chunks of a row XOR-swizzled on the source
//     side: the row-per-lane reads (ds_read_b128) and the column reads (ds_read_b32) of one tile are both conflict free;
//   * the mask is ONE bit (or byte) per (frame, query, key), shared by the heads (the reference materialises 8 byte copies):
//     bit-packed rows come from the mask kernel (attnmask.hip), byte rows (pitch % 4 == 0) are packed per wave.
// Backward, two passes that each recompute the scores in the orientation they need (no atomics, no dQ round trip):
//   pass A (a wave owns 32 queries, walks the key tiles): S^T, dP^T = V . dO^T, dS^T, dQ^T += K^T . dS^T;
//   pass B (a wave owns 32 keys, walks the query tiles):   S, dP = dO . V^T, dS, dV^T += dO^T . P, dK^T += Q^T . dS.
// Work decomposition and staging of each kernel: see the comments above attn_fwd_kernel / attn_bwd_dq_kernel / attn_bwd_dkv_kernel.
#include <math.h>
#include <cstdlib>

#include "combo_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int kD = 32;          // head_dim

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
  const unsigned* mbits;      // the same mask bit-packed: [B, Lq, wpitch] words, bit k of word w = key 32 w + k (or nullptr)
  int wpitch;
  int B, H, Lq, Lk;
  float scale;
  float* out;  // forward: [B, Lq, H*32]
  float* lse;  // [B, H, Lq]  log-sum-exp of the scaled, masked scores
  // backward
  const float* dout;   // [B, Lq, H*32]
  const float* delta;  // [B, H, Lq]  sum_d dO * O
  float* dq; float* dk; float* dv;  // [B, Lq, H*32], [B, Lk, lddk], [B, Lk, lddv]
  long long lddk, lddv;             // row pitch (floats) of dk / dv: H*32, or the pitch of a wider buffer whose column block they are
  unsigned long long* ts;
  int dbg;  // COMBO_ATTN_DBG ablation bits (timing experiments only): 1 no DMA in the loop, 2 no softmax, 4 no MFMA, 8 no mask pack
  int heavy_per_pair, heavy_blocks;  // forward job table: 32-query tiles per pair on the matrix cores; blocks of those jobs
};

// 16 consecutive channels (half g) of one row as 4 float4
__device__ __forceinline__ void load_row16(const float* row, int g, float (&f)[16], float mul = 1.f) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f4v t = *reinterpret_cast<const f4v*>(row + 16 * g + 4 * j);
    f[4 * j] = t.x * mul; f[4 * j + 1] = t.y * mul; f[4 * j + 2] = t.z * mul; f[4 * j + 3] = t.w * mul;
  }
}

// ------------------------------------------------------------------------------------------------ forward
// One workgroup = 2 waves = one JOB: a 32-query tile of one (frame, head) pair ("heavy", matrix cores) or the <= 4 queries
// that are left when Lq % 32 <= 4 ("light", plain FMAs: 100 queries = 3 heavy tiles + 4 queries; a fourth matrix tile would
// spend 7/8 of its MFMA time on padding).  The two waves take the key tiles of alternating parity and merge their
// (max, sum, O) triples at the end: the hardware interleaves one wave's softmax arithmetic with the other's MFMAs, and
// 960 heavy jobs x 2 waves fill the 1024 SIMDs evenly (one job per 4-wave workgroup and pair left 64 CUs with double load).
//   * K / V tiles (32 keys x 32 channels) are staged by LDS-DMA into wave-private double buffers two tiles ahead - no
//     barriers in the main loop, only counted waits; the 16-byte chunks of a row are XOR-swizzled on the SOURCE side so that
//     both the row-per-lane reads of K (ds_read_b128) and the column reads of V^T (ds_read_b32) are conflict free;
//   * the jobs of a pair run on one XCD (block index -> (xcd, pair, tile)), so K / V reach each L2 once;
//   * the mask arrives as one 32-bit word per (query, key tile): packed once per wave from the byte mask into LDS.
constexpr int kTileBytes = 32 * kD * 4;                       // one staged tile
constexpr int kWaveLds = 4 * kTileBytes;                      // K[2] + V[2]
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

__device__ __forceinline__ void dma16(const float* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// rows row0 .. row0+31 (clamped to Lk - 1) x 32 channels -> buf; LDS slot (row, s) holds the row's 16-byte chunk s ^ (row & 7)
__device__ __forceinline__ void stage_tile(const float* base, long long ld, int row0, int Lk, char* buf, int lane) {
  const int rl = lane >> 3, chunk = (lane & 7) ^ rl;
  if (row0 + 32 <= Lk) {  // (wave-uniform) a tile inside the sequence: scalar tile base + one 32-bit lane offset per instruction
    const float* tb = base + (long long)row0 * ld;
    const unsigned o0 = (unsigned)rl * (unsigned)ld + (unsigned)chunk * 4u, step = 8u * (unsigned)ld;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(tb + (o0 + i * step), buf + i * 1024);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dma16(base + (long long)min(row0 + 8 * i + rl, Lk - 1) * ld + chunk * 4, buf + i * 1024);
  }
}
// 16 consecutive channels (half g) of row c of a staged tile
__device__ __forceinline__ void lds_row16(const char* buf, int c, int g, float (&f)[16]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f4v t = *reinterpret_cast<const f4v*>(buf + c * 128 + (((4 * g + j) ^ (c & 7)) << 4));
    f[4 * j] = t.x; f[4 * j + 1] = t.y; f[4 * j + 2] = t.z; f[4 * j + 3] = t.w;
  }
}
// channel c of the 16 rows row_of(e, g) of a staged tile
__device__ __forceinline__ void lds_col16(const char* buf, int c, int g, float (&f)[16]) {
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int r = row_of(e, g);
    f[e] = *reinterpret_cast<const float*>(buf + r * 128 + (((c >> 2) ^ (r & 7)) << 4) + (c & 3) * 4);
  }
}
// 32 blocked flags (bytes at mrow + col0 ..) -> one word; the reads are clamped to the row, the caller masks keys >= Lk
__device__ __forceinline__ unsigned pack_mask_word(const unsigned char* mrow, int pitch, int col0) {
  unsigned w = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned d = *reinterpret_cast<const unsigned*>(mrow + min(col0 + 4 * j, pitch - 4)) & 0x01010101u;
    w |= ((d * 0x00204081u) >> 21 & 0xFu) << (4 * j);
  }
  return w;
}

// A workgroup's job (shared by the forward kernel and backward pass A): which queries, this wave's staging buffers, the first
// two K / V tiles on their way and the mask words of this wave's key tiles in LDS.
struct Job {
  int b, h, q0, nq, n_my, mstride;
  bool light, valid;
  char* kbuf; char* vbuf;
  unsigned* mlds;
  const float* kb; const float* vb;
};
__device__ __forceinline__ Job job_open(const AttnArgs& a, char* smem, int lane, int wave) {
  Job j;
  const int n_kt = (a.Lk + 31) / 32;
  int pair;
  j.light = false;
  if ((int)blockIdx.x < a.heavy_blocks) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pl = slot / a.heavy_per_pair;
    pair = pl * 8 + xcd;
    j.q0 = (slot - pl * a.heavy_per_pair) * 32;
    j.nq = min(32, a.Lq - j.q0);
  } else {
    pair = blockIdx.x - a.heavy_blocks;
    j.q0 = a.heavy_per_pair * 32;
    j.nq = a.Lq - j.q0;
    j.light = true;
  }
  j.valid = pair < a.B * a.H;
  if (!j.valid) return j;
  j.b = pair / a.H;
  j.h = pair - j.b * a.H;
  j.kbuf = smem + wave * kWaveLds;
  j.vbuf = j.kbuf + 2 * kTileBytes;
  j.mstride = ((n_kt + 1) >> 1) | 1;  // words per query row in LDS: this wave's tiles, odd pitch
  j.mlds = reinterpret_cast<unsigned*>(smem + 2 * kWaveLds) + wave * 32 * j.mstride;
  j.kb = a.k + (long long)j.b * a.Lk * a.ldk + j.h * kD;
  j.vb = a.v + (long long)j.b * a.Lk * a.ldv + j.h * kD;
  j.n_my = (a.dbg & 32) ? 0 : (n_kt - wave + 1) >> 1;  // this wave's tiles: wave, wave + 2, ..
  // operand tiles 0 and 1 of this wave on their way (always 8 DMA instructions per tile: the waits in the loops count them)
  if (!(a.dbg & 64)) {
    stage_tile(j.kb, a.ldk, wave * 32, a.Lk, j.kbuf, lane);
    stage_tile(j.vb, a.ldv, wave * 32, a.Lk, j.vbuf, lane);
    stage_tile(j.kb, a.ldk, (wave + 2) * 32, a.Lk, j.kbuf + kTileBytes, lane);
    stage_tile(j.vb, a.ldv, (wave + 2) * 32, a.Lk, j.vbuf + kTileBytes, lane);
  }
  // mask words of this wave's tiles
  for (int idx = lane; idx < 32 * j.n_my; idx += 64) {
    const int r = idx / j.n_my, i = idx - r * j.n_my;
    const int kt = wave + 2 * i;
    const long long row = (long long)j.b * a.Lq + min(j.q0 + r, a.Lq - 1);
    unsigned w = 0;
    if (a.mbits) w = a.mbits[row * a.wpitch + kt];
    else if (a.mask && !(a.dbg & 8)) w = pack_mask_word(a.mask + row * a.pitch, a.pitch, kt * 32);
    if (kt * 32 + 32 > a.Lk) w |= ~0u << (a.Lk - kt * 32);  // keys beyond Lk (last tile only; Lk - kt*32 in 1..31)
    j.mlds[r * j.mstride + i] = w;
  }
  return j;
}

__global__ void __launch_bounds__(128, 2)
attn_fwd_kernel(const AttnArgs a) {
  combo_ts_begin(a.ts);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, g = lane >> 5;
  const Job jb = job_open(a, smem, lane, wave);
  if (jb.valid) {
    const int b = jb.b, h = jb.h, q0 = jb.q0, nq = jb.nq, n_my = jb.n_my, mstride = jb.mstride;
    const bool light = jb.light;
    char* kbuf = jb.kbuf;
    char* vbuf = jb.vbuf;
    unsigned* mlds = jb.mlds;
    const float* kb = jb.kb;
    const float* vb = jb.vb;
    float m = -1e30f, l = 0.f;  // running max (log2 units) and sum of this wave's keys
    if (!light) {
      // ---------------------------------------------------------------- heavy: S^T = K . Q^T on the matrix cores
      const int qc = min(q0 + c, a.Lq - 1);
      float qf[16];
      load_row16(a.q + ((long long)b * a.Lq + qc) * a.ldq + h * kD, g, qf, a.scale * kLog2e);
      f32x16 o;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[e] = 0.f;
      for (int i = 0; i < n_my; ++i) {
        const char* kt_buf = kbuf + (i & 1) * kTileBytes;
        const char* vt_buf = vbuf + (i & 1) * kTileBytes;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile i landed (tile i + 1 may still be in flight)
        __builtin_amdgcn_sched_barrier(0);
        float kf[16], vf[16];
        lds_row16(kt_buf, c, g, kf);
        lds_col16(vt_buf, c, g, vf);
        const unsigned mw = mlds[c * mstride + i] >> (4 * g);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!(a.dbg & 1)) {
          stage_tile(kb, a.ldk, (wave + 2 * (i + 2)) * 32, a.Lk, const_cast<char*>(kt_buf), lane);
          stage_tile(vb, a.ldv, (wave + 2 * (i + 2)) * 32, a.Lk, const_cast<char*>(vt_buf), lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
        if (!(a.dbg & 4)) {
#pragma unroll
          for (int t = 0; t < 16; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t], qf[t], s, 0, 0, 0);
        } else {
#pragma unroll
          for (int t = 0; t < 16; ++t) s[t] = kf[t] * qf[t];
        }
        float mx = -1e30f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          // blocked -> -inf: sign-extended bit (0 / ~0) selects between the score and 0xff800000 (v_bfe_i32 + v_bfi_b32)
          const unsigned sel = (unsigned)__builtin_amdgcn_sbfe(mw, (e & 3) + 8 * (e >> 2), 1);
          s[e] = __uint_as_float((__float_as_uint(s[e]) & ~sel) | (0xff800000u & sel));
          mx = fmaxf(mx, s[e]);
        }
        mx = fmaxf(mx, xor32(mx));
        const float m_new = fmaxf(m, mx);  // finite: a blocked cell gives exp2(-inf - m_new) = 0 without a select
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        m = m_new;
        float ps = 0.f;
        float p[16];
        if (!(a.dbg & 2)) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            p[e] = __builtin_amdgcn_exp2f(s[e] - m_new);
            ps += p[e];
            o[e] *= alpha;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) p[e] = s[e];
        }
        l = l * alpha + ps;
        if (!(a.dbg & 4)) {
#pragma unroll
          for (int e = 0; e < 16; ++e) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[e], p[e], o, 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) o[e] += vf[e] * p[e];
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two speculative tiles: nothing may land after the buffers are reused
      l += xor32(l);
      // merge the two waves: wave 1 hands (m, l, O) over through its staging buffers
      float* xch = reinterpret_cast<float*>(smem + kWaveLds);
      __syncthreads();
      if (wave == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) xch[e * 64 + lane] = o[e];
        xch[16 * 64 + lane] = m;
        xch[17 * 64 + lane] = l;
      }
      __syncthreads();
      if (wave == 0) {
        const float m1 = xch[16 * 64 + lane], l1 = xch[17 * 64 + lane];
        const float mm = fmaxf(m, m1);
        const float a0 = __builtin_amdgcn_exp2f(m - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);
        const float lt = l * a0 + l1 * a1;
        const float inv = lt > 0.f ? 1.f / lt : 0.f;
        if (c < nq) {
          float* orow = a.out + ((long long)b * a.Lq + q0 + c) * (a.H * kD) + h * kD;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f4v r;
#pragma unroll
            for (int x = 0; x < 4; ++x) r[x] = (o[4 * j + x] * a0 + xch[(4 * j + x) * 64 + lane] * a1) * inv;
            *reinterpret_cast<f4v*>(orow + 8 * j + 4 * g) = r;
          }
          if (g == 0 && a.lse) a.lse[((long long)b * a.H + h) * a.Lq + q0 + c] = (mm + log2f(lt)) * kLn2;
        }
      }
    } else {
      // ---------------------------------------------------------------- light: <= 4 queries, plain FMAs
      // lane = (key c of the tile, channel half g): every lane runs its OWN online softmax over its keys (13 per query at
      // 784 keys) and accumulates O[query][its 16 channels] - nothing crosses lanes inside the loop except the two halves of a
      // dot product; the 32 key lanes (and the two waves) are merged once at the end.
      float o4[4][16], m4[4], l4[4];
#pragma unroll
      for (int qi = 0; qi < 4; ++qi) {
        m4[qi] = -1e30f;
        l4[qi] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) o4[qi][t] = 0.f;
      }
      float* ql = reinterpret_cast<float*>(smem + 2 * kWaveLds + 2 * 32 * mstride * 4) + wave * 264;  // q[4][32], pre-scaled
      for (int idx = lane; idx < 128; idx += 64)
        ql[idx] = a.q[((long long)b * a.Lq + min(q0 + (idx >> 5), a.Lq - 1)) * a.ldq + h * kD + (idx & 31)] * (a.scale * kLog2e);
      for (int i = 0; i < n_my; ++i) {
        const char* kt_buf = kbuf + (i & 1) * kTileBytes;
        const char* vt_buf = vbuf + (i & 1) * kTileBytes;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        float kf[16], vr[16];
        lds_row16(kt_buf, c, g, kf);
        lds_row16(vt_buf, c, g, vr);
        unsigned mw[4];
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) mw[qi] = mlds[qi * mstride + i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        stage_tile(kb, a.ldk, (wave + 2 * (i + 2)) * 32, a.Lk, const_cast<char*>(kt_buf), lane);
        stage_tile(vb, a.ldv, (wave + 2 * (i + 2)) * 32, a.Lk, const_cast<char*>(vt_buf), lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
          float sp = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f4v qq = *reinterpret_cast<const f4v*>(ql + qi * 32 + 16 * g + 4 * j);  // one address per half: broadcast
#pragma unroll
            for (int x = 0; x < 4; ++x) sp = fmaf(kf[4 * j + x], qq[x], sp);
          }
          float sv = sp + xor32(sp);
          sv = ((mw[qi] >> c) & 1u) ? -INFINITY : sv;
          const float m_new = fmaxf(m4[qi], sv);
          const float alpha = __builtin_amdgcn_exp2f(m4[qi] - m_new);
          const float pe = __builtin_amdgcn_exp2f(sv - m_new);
          m4[qi] = m_new;
          l4[qi] = l4[qi] * alpha + pe;
#pragma unroll
          for (int t = 0; t < 16; ++t) o4[qi][t] = fmaf(o4[qi][t], alpha, pe * vr[t]);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // merge the 32 key lanes of each half: common maximum, then sums over the lanes through the (now idle) staging buffers
      float* tb = reinterpret_cast<float*>(kbuf);  // [128 rows (query, channel)][32 key lanes], this wave's 16 KB
#pragma unroll
      for (int qi = 0; qi < 4; ++qi) {
        float mm = m4[qi];
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) mm = fmaxf(mm, __shfl_xor(mm, d, 64));
        const float sc = __builtin_amdgcn_exp2f(m4[qi] - mm);
        m4[qi] = mm;
        l4[qi] *= sc;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) l4[qi] += __shfl_xor(l4[qi], d, 64);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int r = qi * 32 + 16 * g + t;
          tb[r * 32 + (c ^ (r & 31))] = o4[qi][t] * sc;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float osum[2];  // rows lane and lane + 64: (query lane >> 5, channel lane & 31) and (query (lane >> 5) + 2, same channel)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f4v t4 = *reinterpret_cast<const f4v*>(tb + (lane + 64 * k) * 32 + ((4 * j + 4 * lane) & 31));
          acc += (t4.x + t4.y) + (t4.z + t4.w);
        }
        osum[k] = acc;
      }
      // the two waves: wave 1 hands (max, sum, O) over
      float* xch = reinterpret_cast<float*>(smem + kWaveLds);
      __syncthreads();
      if (wave == 1) {
        xch[lane] = osum[0];
        xch[64 + lane] = osum[1];
        if (lane < 4) {
          float mv = m4[0], lv = l4[0];
#pragma unroll
          for (int qi = 1; qi < 4; ++qi)
            if (lane == qi) { mv = m4[qi]; lv = l4[qi]; }
          xch[128 + lane] = mv;
          xch[132 + lane] = lv;
        }
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int qi = (lane >> 5) + 2 * k;
          float m0 = m4[0], l0 = l4[0];
#pragma unroll
          for (int x = 1; x < 4; ++x)
            if (qi == x) { m0 = m4[x]; l0 = l4[x]; }
          const float m1 = xch[128 + qi], l1 = xch[132 + qi];
          const float mm = fmaxf(m0, m1);
          const float a0 = __builtin_amdgcn_exp2f(m0 - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);
          const float lt = l0 * a0 + l1 * a1;
          const float inv = lt > 0.f ? 1.f / lt : 0.f;
          if (qi < nq) {
            a.out[((long long)b * a.Lq + q0 + qi) * (a.H * kD) + h * kD + c] = (osum[k] * a0 + xch[64 * k + lane] * a1) * inv;
            if (c == 0 && a.lse) a.lse[((long long)b * a.H + h) * a.Lq + q0 + qi] = (mm + log2f(lt)) * kLn2;
          }
        }
      }
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
// The forward kernel's jobs and staging (2 waves per 32-query tile, alternate key tiles, K / V tiles through LDS-DMA double
// buffers, the <= 4 left-over queries on plain FMAs); per tile S^T = K . Q^T, dP^T = V . dO^T, dS^T = P o (dP - delta),
// dQ^T += K^T . dS^T (K^T is the column read of the tile that S^T reads by rows).  The two waves' partial dQ are added at the end.
__global__ void __launch_bounds__(128, 2)
attn_bwd_dq_kernel(const AttnArgs a) {
  combo_ts_begin(a.ts);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, g = lane >> 5;
  const Job jb = job_open(a, smem, lane, wave);
  if (jb.valid) {
    const int b = jb.b, h = jb.h, q0 = jb.q0, nq = jb.nq, n_my = jb.n_my, mstride = jb.mstride;
    char* kbuf = jb.kbuf;
    char* vbuf = jb.vbuf;
    const unsigned* mlds = jb.mlds;
    const float* kb = jb.kb;
    const float* vb = jb.vb;
    const float* lse_p = a.lse + ((long long)b * a.H + h) * a.Lq;
    const float* dl_p = a.delta + ((long long)b * a.H + h) * a.Lq;
    if (!jb.light) {
      const int qc = min(q0 + c, a.Lq - 1);
      float qf[16], dof[16];
      load_row16(a.q + ((long long)b * a.Lq + qc) * a.ldq + h * kD, g, qf, a.scale * kLog2e);
      load_row16(a.dout + ((long long)b * a.Lq + qc) * (a.H * kD) + h * kD, g, dof);
      const float lse2 = lse_p[qc] * kLog2e, dl = dl_p[qc];
      f32x16 dq;
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[e] = 0.f;
      for (int i = 0; i < n_my; ++i) {
        const char* kt_buf = kbuf + (i & 1) * kTileBytes;
        const char* vt_buf = vbuf + (i & 1) * kTileBytes;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        float kf[16], vr[16], kcol[16];
        lds_row16(kt_buf, c, g, kf);
        lds_row16(vt_buf, c, g, vr);
        lds_col16(kt_buf, c, g, kcol);
        const unsigned mw = mlds[c * mstride + i] >> (4 * g);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        stage_tile(kb, a.ldk, (wave + 2 * (i + 2)) * 32, a.Lk, const_cast<char*>(kt_buf), lane);
        stage_tile(vb, a.ldv, (wave + 2 * (i + 2)) * 32, a.Lk, const_cast<char*>(vt_buf), lane);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 s, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
        for (int t = 0; t < 16; ++t) {  // two independent accumulation chains, interleaved
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t], qf[t], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[t], dof[t], dp, 0, 0, 0);
        }
        float ds[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned sel = (unsigned)__builtin_amdgcn_sbfe(mw, (e & 3) + 8 * (e >> 2), 1);  // ~0 for a blocked cell
          const float pe = __builtin_amdgcn_exp2f(s[e] - lse2) * (dp[e] - dl);
          ds[e] = __uint_as_float(__float_as_uint(pe) & ~sel);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) dq = __builtin_amdgcn_mfma_f32_32x32x2f32(kcol[e], ds[e], dq, 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      float* xch = reinterpret_cast<float*>(smem + kWaveLds);
      __syncthreads();
      if (wave == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) xch[e * 64 + lane] = dq[e];
      }
      __syncthreads();
      if (wave == 0 && c < nq) {
        float* row = a.dq + ((long long)b * a.Lq + q0 + c) * (a.H * kD) + h * kD;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f4v r;
#pragma unroll
          for (int x = 0; x < 4; ++x) r[x] = (dq[4 * j + x] + xch[(4 * j + x) * 64 + lane]) * a.scale;
          *reinterpret_cast<f4v*>(row + 8 * j + 4 * g) = r;
        }
      }
    } else {
      // ---------------------------------------------------------------- light: <= 4 queries, plain FMAs; lane = (key c, half g)
      float dq4[4][16];
#pragma unroll
      for (int qi = 0; qi < 4; ++qi)
#pragma unroll
        for (int t = 0; t < 16; ++t) dq4[qi][t] = 0.f;
      float* ql = reinterpret_cast<float*>(smem + 2 * kWaveLds + 2 * 32 * mstride * 4) + wave * 264;  // q[4][32], dO[4][32], lse[4], delta[4]
      float* dol = ql + 128;
      for (int idx = lane; idx < 128; idx += 64) {
        const long long row = (long long)b * a.Lq + min(q0 + (idx >> 5), a.Lq - 1);
        ql[idx] = a.q[row * a.ldq + h * kD + (idx & 31)] * (a.scale * kLog2e);
        dol[idx] = a.dout[row * (a.H * kD) + h * kD + (idx & 31)];
      }
      if (lane < 4) {
        ql[256 + lane] = lse_p[min(q0 + lane, a.Lq - 1)] * kLog2e;
        ql[260 + lane] = dl_p[min(q0 + lane, a.Lq - 1)];
      }
      for (int i = 0; i < n_my; ++i) {
        const char* kt_buf = kbuf + (i & 1) * kTileBytes;
        const char* vt_buf = vbuf + (i & 1) * kTileBytes;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        float kf[16], vr[16];
        lds_row16(kt_buf, c, g, kf);
        lds_row16(vt_buf, c, g, vr);
        unsigned mw[4];
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) mw[qi] = mlds[qi * mstride + i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        stage_tile(kb, a.ldk, (wave + 2 * (i + 2)) * 32, a.Lk, const_cast<char*>(kt_buf), lane);
        stage_tile(vb, a.ldv, (wave + 2 * (i + 2)) * 32, a.Lk, const_cast<char*>(vt_buf), lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
          float sp = 0.f, dpp = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f4v qq = *reinterpret_cast<const f4v*>(ql + qi * 32 + 16 * g + 4 * j);
            const f4v dd = *reinterpret_cast<const f4v*>(dol + qi * 32 + 16 * g + 4 * j);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              sp = fmaf(kf[4 * j + x], qq[x], sp);
              dpp = fmaf(vr[4 * j + x], dd[x], dpp);
            }
          }
          const float sv = sp + xor32(sp), dpv = dpp + xor32(dpp);
          const float pe = ((mw[qi] >> c) & 1u) ? 0.f : __builtin_amdgcn_exp2f(sv - ql[256 + qi]);
          const float dsv = pe * (dpv - ql[260 + qi]);
#pragma unroll
          for (int t = 0; t < 16; ++t) dq4[qi][t] = fmaf(dsv, kf[t], dq4[qi][t]);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // sum over the 32 key lanes through the (idle) staging buffers, then over the two waves
      float* tb = reinterpret_cast<float*>(kbuf);
#pragma unroll
      for (int qi = 0; qi < 4; ++qi)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int r = qi * 32 + 16 * g + t;
          tb[r * 32 + (c ^ (r & 31))] = dq4[qi][t];
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float osum[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f4v t4 = *reinterpret_cast<const f4v*>(tb + (lane + 64 * k) * 32 + ((4 * j + 4 * lane) & 31));
          acc += (t4.x + t4.y) + (t4.z + t4.w);
        }
        osum[k] = acc;
      }
      float* xch = reinterpret_cast<float*>(smem + kWaveLds);
      __syncthreads();
      if (wave == 1) { xch[lane] = osum[0]; xch[64 + lane] = osum[1]; }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int qi = (lane >> 5) + 2 * k;
          if (qi < nq) a.dq[((long long)b * a.Lq + q0 + qi) * (a.H * kD) + h * kD + c] = (osum[k] + xch[64 * k + lane]) * a.scale;
        }
      }
    }
  }
  combo_ts_end(a.ts);
}

// ------------------------------------------------------------------------------------------------ backward, pass B: dK, dV
// One workgroup = 4 waves = 4 key tiles of one (frame, head) pair; a wave owns 32 keys (K / V rows in registers as B
// operands) and walks the query tiles.  The Q and dO tiles of up to 128 queries (all of them for the decoder's 100) are
// staged ONCE per workgroup by LDS-DMA (same swizzle as the forward tiles) and read by all four waves, by rows (S = Q . K^T,
// dP = dO . V^T) and by columns (dV^T += dO^T . P, dK^T += Q^T . dS); lse / delta and the mask words of the 4 key tiles sit
// next to them, laid out so that the 4 query rows of a register group come with one 16-byte read.
constexpr int kQChunk = 128;  // queries per LDS fill
__global__ void __launch_bounds__(256, 3)
attn_bwd_dkv_kernel(const AttnArgs a) {
  combo_ts_begin(a.ts);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int pair = blockIdx.x, b = pair / a.H, h = pair - b * a.H;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, g = lane >> 5;  // here the column index is the KEY
  const int n_kt = (a.Lk + 31) / 32;
  const int kt = min((int)blockIdx.y * 4 + wave, n_kt - 1);  // (a surplus wave repeats the last tile and does not store)
  const bool tile_ok = (int)blockIdx.y * 4 + wave < n_kt;
  char* qt_lds = smem;                       // [4 tiles][4 KB]
  char* do_lds = smem + 4 * kTileBytes;      // [4 tiles][4 KB]
  float* lse_lds = reinterpret_cast<float*>(smem + 8 * kTileBytes);  // [128] log2-scaled
  float* dl_lds = lse_lds + kQChunk;                                  // [128]
  unsigned* mw_lds = reinterpret_cast<unsigned*>(dl_lds + kQChunk);   // [4 key tiles][128 queries]
  const int ki = kt * 32 + c;
  const bool k_ok = tile_ok && ki < a.Lk;
  const int kc = min(ki, a.Lk - 1);
  float kcol[16], vcol[16];
  load_row16(a.k + ((long long)b * a.Lk + kc) * a.ldk + h * kD, g, kcol);
  load_row16(a.v + ((long long)b * a.Lk + kc) * a.ldv + h * kD, g, vcol);
  const float* qb = a.q + (long long)b * a.Lq * a.ldq + h * kD;
  const float* dob = a.dout + (long long)b * a.Lq * (a.H * kD) + h * kD;
  const float* lseb = a.lse + ((long long)b * a.H + h) * a.Lq;
  const float* dlb = a.delta + ((long long)b * a.H + h) * a.Lq;
  const float sc2 = a.scale * kLog2e;
  f32x16 dk, dv;
#pragma unroll
  for (int e = 0; e < 16; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
  for (int qbase = 0; qbase < a.Lq; qbase += kQChunk) {
    if (qbase) __syncthreads();  // everyone is done with the previous fill
    // wave w stages query tile w of the chunk (Q and dO); rows beyond Lq repeat the last row (their mask words are all ones)
    stage_tile(qb, a.ldq, qbase + wave * 32, a.Lq, qt_lds + wave * kTileBytes, lane);
    stage_tile(dob, (long long)a.H * kD, qbase + wave * 32, a.Lq, do_lds + wave * kTileBytes, lane);
    if (threadIdx.x < kQChunk) {
      const int q = qbase + threadIdx.x;
      lse_lds[threadIdx.x] = q < a.Lq ? lseb[q] * kLog2e : 0.f;
      dl_lds[threadIdx.x] = q < a.Lq ? dlb[q] : 0.f;
    }
    for (int idx = threadIdx.x; idx < 4 * kQChunk; idx += 256) {
      const int t = idx >> 7, ql = idx & (kQChunk - 1);
      const int q = qbase + ql, ktw = min((int)blockIdx.y * 4 + t, n_kt - 1);
      unsigned w = 0;
      if (q >= a.Lq) w = ~0u;
      else if (a.mbits) w = a.mbits[((long long)b * a.Lq + q) * a.wpitch + ktw];
      else if (a.mask) w = pack_mask_word(a.mask + ((long long)b * a.Lq + q) * a.pitch, a.pitch, ktw * 32);
      mw_lds[idx] = w;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int n_qt = min(4, (a.Lq - qbase + 31) / 32);
    for (int qt = 0; qt < n_qt; ++qt) {
      const char* qtile = qt_lds + qt * kTileBytes;
      const char* dtile = do_lds + qt * kTileBytes;
      float qrow[16], dorow[16], qcol[16], docol[16];
      lds_row16(qtile, c, g, qrow);
      lds_row16(dtile, c, g, dorow);
      lds_col16(qtile, c, g, qcol);
      lds_col16(dtile, c, g, docol);
      f4v lse4[4], dl4[4];
      uint4 mw4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // rows 8 j + 4 g + {0..3} of the tile: one register group
        lse4[j] = *reinterpret_cast<const f4v*>(lse_lds + qt * 32 + 8 * j + 4 * g);
        dl4[j] = *reinterpret_cast<const f4v*>(dl_lds + qt * 32 + 8 * j + 4 * g);
        mw4[j] = *reinterpret_cast<const uint4*>(mw_lds + wave * kQChunk + qt * 32 + 8 * j + 4 * g);
      }
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[t], kcol[t], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dorow[t], vcol[t], dp, 0, 0, 0);
      }
      float p[16], ds[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const unsigned w = (e & 3) == 0 ? mw4[e >> 2].x : (e & 3) == 1 ? mw4[e >> 2].y : (e & 3) == 2 ? mw4[e >> 2].z : mw4[e >> 2].w;
        const unsigned sel = 0u - ((w >> c) & 1u);  // ~0 for a blocked cell
        const float pe = __builtin_amdgcn_exp2f(fmaf(s[e], sc2, -lse4[e >> 2][e & 3]));
        p[e] = __uint_as_float(__float_as_uint(pe) & ~sel);
        ds[e] = p[e] * (dp[e] - dl4[e >> 2][e & 3]);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        dv = __builtin_amdgcn_mfma_f32_32x32x2f32(docol[e], p[e], dv, 0, 0, 0);
        dk = __builtin_amdgcn_mfma_f32_32x32x2f32(qcol[e], ds[e], dk, 0, 0, 0);
      }
    }
  }
  if (k_ok) {
    float* rk = a.dk + ((long long)b * a.Lk + ki) * a.lddk + h * kD;
    float* rv = a.dv + ((long long)b * a.Lk + ki) * a.lddv + h * kD;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<f4v*>(rk + 8 * j + 4 * g) = f4v{dk[4 * j] * a.scale, dk[4 * j + 1] * a.scale, dk[4 * j + 2] * a.scale, dk[4 * j + 3] * a.scale};
      *reinterpret_cast<f4v*>(rv + 8 * j + 4 * g) = f4v{dv[4 * j], dv[4 * j + 1], dv[4 * j + 2], dv[4 * j + 3]};
    }
  }
  combo_ts_end(a.ts);
}

bool common_ok(const float* q, long long ldq, const float* k, long long ldk, const float* v, long long ldv, int B, int H, int Lq, int Lk,
               const unsigned char* mask, int pitch, const unsigned* bits = nullptr, int wpitch = 0) {
  if (bits && (wpitch < (Lk + 31) / 32 || ((uintptr_t)bits & 3))) return false;
  return q && k && v && B > 0 && H > 0 && Lq > 0 && Lk > 0 && ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 &&
         !((uintptr_t)q & 15) && !((uintptr_t)k & 15) && !((uintptr_t)v & 15) &&
         (!mask || (pitch % 4 == 0 && pitch >= Lk && pitch >= 4 && !((uintptr_t)mask & 3)));
}

// job table of the query-major kernels (forward, backward pass A) -> dynamic LDS bytes and grid size
bool job_table(AttnArgs& a, int& lds, int& grid) {
  static const int dbg = [] { const char* e = getenv("COMBO_ATTN_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
  const int rem = a.Lq % 32;
  const bool light = rem > 0 && rem <= 4 && !(dbg & 16);
  a.heavy_per_pair = a.Lq / 32 + (rem > 4 ? 1 : 0);
  a.heavy_blocks = (a.B * a.H + 7) / 8 * 8 * a.heavy_per_pair;
  const int n_kt = (a.Lk + 31) / 32;
  const int mstride = ((n_kt + 1) >> 1) | 1;
  lds = 2 * kWaveLds + 2 * 32 * mstride * 4 + 2 * 264 * 4;
  grid = a.heavy_blocks + (light ? a.B * a.H : 0);
  return lds <= 160 * 1024;
}

}  // namespace

extern "C" int combo_attention_forward_f32(const float* q, long long ldq, const float* k, long long ldk, const float* v, long long ldv,
                                           const unsigned char* blocked, int pitch, const unsigned* blocked_bits, int wpitch, int B,
                                           int H, int Lq, int Lk, float scale, float* out, float* lse, combo_stream_t stream) {
  if (!common_ok(q, ldq, k, ldk, v, ldv, B, H, Lq, Lk, blocked, pitch, blocked_bits, wpitch) || !out || ((uintptr_t)out & 15))
    return COMBO_EINVAL;
  AttnArgs a{q, k, v, ldq, ldk, ldv, blocked, pitch, blocked_bits, wpitch, B, H, Lq, Lk, scale, out, lse, nullptr, nullptr, nullptr, nullptr, nullptr,
             0, 0, combo_timing_next_slot(COMBO_TS_ATTN_FWD, 4.0 * B * H * (double)Lq * Lk * kD)};
  int lds = 0, grid = 0;
  if (!job_table(a, lds, grid)) return COMBO_EINVAL;
  static int lds_set = 0;
  if (lds > lds_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    lds_set = lds;
  }
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(grid), dim3(128), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

/* _ld: dk / dv are written with row pitches lddk / lddv (floats, multiples of 4, >= H*32): the K / V gradients of a layer land in
 * their column block of one [B*Lk, layers*E] buffer per memory level, which the merged K / V projection of that level
 * (ops/linear.py memory_kv) contracts in ONE input-gradient GEMM - no concatenation, no accumulation adds. */
extern "C" int combo_attention_backward_ld_f32(const float* q, long long ldq, const float* k, long long ldk, const float* v,
                                               long long ldv, const unsigned char* blocked, int pitch, const unsigned* blocked_bits,
                                               int wpitch, int B, int H, int Lq, int Lk, float scale, const float* out,
                                               const float* lse, const float* dout, float* delta_ws, float* dq, float* dk,
                                               long long lddk, float* dv, long long lddv, combo_stream_t stream) {
  if (!common_ok(q, ldq, k, ldk, v, ldv, B, H, Lq, Lk, blocked, pitch, blocked_bits, wpitch) || !out || !lse || !dout || !delta_ws || !dq || !dk || !dv ||
      (((uintptr_t)out | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) || lddk < H * kD || lddv < H * kD ||
      lddk % 4 != 0 || lddv % 4 != 0)
    return COMBO_EINVAL;
  const long long n = (long long)B * Lq * H;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, out, B, H, Lq, delta_ws);
  AttnArgs a{q, k, v, ldq, ldk, ldv, blocked, pitch, blocked_bits, wpitch, B, H, Lq, Lk, scale, nullptr, const_cast<float*>(lse), dout, delta_ws, dq, dk, dv,
             lddk, lddv, combo_timing_next_slot(COMBO_TS_ATTN_BWD, 6.0 * B * H * (double)Lq * Lk * kD)};
  {
    int lds = 0, grid = 0;
    if (!job_table(a, lds, grid)) return COMBO_EINVAL;
    static int lds_set = 0;
    if (lds > lds_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return (int)e;
      lds_set = lds;
    }
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(grid), dim3(128), lds, (hipStream_t)stream, a);
  }
  a.ts = combo_timing_next_slot(COMBO_TS_ATTN_BWD, 8.0 * B * H * (double)Lq * Lk * kD);
  const int n_kt = (Lk + 31) / 32;
  const int lds_kv = 8 * kTileBytes + 2 * kQChunk * 4 + 4 * kQChunk * 4;
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(B * H, (n_kt + 3) / 4), dim3(256), lds_kv, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" int combo_attention_backward_f32(const float* q, long long ldq, const float* k, long long ldk, const float* v, long long ldv,
                                            const unsigned char* blocked, int pitch, const unsigned* blocked_bits, int wpitch, int B,
                                            int H, int Lq, int Lk, float scale, const float* out, const float* lse, const float* dout,
                                            float* delta_ws, float* dq, float* dk, float* dv, combo_stream_t stream) {
  return combo_attention_backward_ld_f32(q, ldq, k, ldk, v, ldv, blocked, pitch, blocked_bits, wpitch, B, H, Lq, Lk, scale, out, lse, dout,
                                         delta_ws, dq, dk, (long long)H * kD, dv, (long long)H * kD, stream);
}
