// Fused `mask_embed @ pixel_embed -> sigmoid < 0.5 -> attention mask` of the decoder's prediction heads.
//
// Reference (transformer_decoder.py:498-507, :458): every prediction head computes the FULL-resolution mask logits
// einsum("bqc,bchw->bqhw"), interpolates them bilinearly to the next layer's memory size (7^2 / 14^2 / 28^2 at 224 x 224),
// thresholds `sigmoid < 0.5`, replicates the bool mask over the 8 heads, and the next layer un-blocks fully blocked rows
// (nonzero() + index_put: a host sync).
//
// Bilinear interpolation and the contraction over the channels are both linear, so they commute:
//     down(mask_embed . mask_features^T) == mask_embed . down(mask_features)^T
// The mask of a layer therefore needs only the DOWNSAMPLED pixel embedding of its level - computed once per forward for the
// three levels (combo_downsample_tokens_f32), the mask features are the same for all 10 heads - and a [Q x hw] contraction,
// 4 .. 64 x smaller than the full one.  combo_mask_bits_f32 does that contraction on the fp32 matrix instruction
// (v_mfma_f32_32x32x2_f32: exact fp32, the threshold sits at 0) and never writes the scores: in the MFMA result layout a lane
// holds 16 queries of ONE key cell, so a wave ballot of `sigmoid(score) < 0.5` IS the bit-packed mask word (bit k of word j
// = key 32 j + k) the attention kernels read (csrc/attention.hip).  The fully-blocked-row reset is decided in the same launch
// (the 4 waves of a workgroup own all key tiles of 32 queries of one frame).  The full-resolution logits - needed by the
// losses only - leave the layer loop: ONE batched GEMM for all 10 heads after it (ops/masklogit.py).
// The reorder moves a score by fp32 round-off only; a cell flips only if its score is within ~1e-6 of 0 (tests bound it).
#include <stdint.h>

#include "combo_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// ATen upsample_bilinear2d source index, align_corners = False (area_pixel_compute_source_index; src < 0 clamps to 0)
__device__ __forceinline__ void src_index(int o, float scale, int in_size, int& i0, int& ip, float& l0, float& l1) {
  float f = scale * (o + 0.5f) - 0.5f;
  f = f < 0.f ? 0.f : f;
  i0 = (int)f;
  ip = (i0 < in_size - 1) ? 1 : 0;
  l1 = f - i0;
  l0 = 1.f - l1;
}

// out[b, oy*w + ox, c] = bilinear(x[b, :, c] viewed [H, W]) ; C % 4 == 0, one lane per 4 channels
__global__ void __launch_bounds__(256)
downsample_tokens_kernel(const float* __restrict__ x, int B, int H, int W, int h, int w, int C, float* __restrict__ out) {
  const long long n = (long long)B * h * w * (C / 4);
  const float sh = (float)H / (float)h, sw = (float)W / (float)w;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % (C / 4));
    const long long cell = i / (C / 4);
    const int ox = (int)(cell % w), oy = (int)((cell / w) % h);
    const int b = (int)(cell / ((long long)w * h));
    int y0, yp, x0, xp;
    float ly0, ly1, lx0, lx1;
    src_index(oy, sh, H, y0, yp, ly0, ly1);
    src_index(ox, sw, W, x0, xp, lx0, lx1);
    const float* p = x + (((long long)b * H + y0) * W + x0) * C + c4 * 4;
    const float4 v00 = *reinterpret_cast<const float4*>(p);
    const float4 v01 = *reinterpret_cast<const float4*>(p + (long long)xp * C);
    const float4 v10 = *reinterpret_cast<const float4*>(p + (long long)yp * W * C);
    const float4 v11 = *reinterpret_cast<const float4*>(p + ((long long)yp * W + xp) * C);
    float4 r;  // the association of the reference's interpolation: ly0 * (lx0 a + lx1 b) + ly1 * (lx0 c + lx1 d)
    r.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
    r.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
    r.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
    r.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
    *reinterpret_cast<float4*>(out + cell * C + c4 * 4) = r;
  }
}

constexpr int kC = 256;        // mask dimension (MODEL.SEM_SEG_HEAD.MASK_DIM of every shipped config)
constexpr int kChunks = kC / 8;  // 16-byte chunks per lane: lane (r, g) holds k = (2 c + g) * 4 .. + 3, c = 0 .. 31
constexpr int kMaxTiles = 128;   // key tiles of 32 cells: hw <= 4096

// One workgroup = (frame b, block of 32 queries); wave w owns the key tiles t = w, w + 4, ...  Scores
// S[q, cell] = sum_c me[b, q, c] * mfd[b, cell, c] on v_mfma_f32_32x32x2_f32 (A = queries from registers, loaded once;
// B = cells streamed from L2 in half-tiles, double buffered).
__global__ void __launch_bounds__(256, 1)
mask_bits_kernel(const float* __restrict__ me, const float* __restrict__ mfd, int B, int Q, int hw, int reset_full_rows,
                 int wpitch, unsigned* __restrict__ bits, int pitch, unsigned char* __restrict__ bytes) {
  __shared__ unsigned words[32][kMaxTiles + 1];  // [query][key tile]
  __shared__ unsigned anyopen[4][32];            // per wave: OR over its tiles of the un-blocked real cells of a query
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, g = lane >> 5;
  const int qb = (Q + 31) / 32;
  const int b = blockIdx.x / qb, q0 = (blockIdx.x % qb) * 32;
  const int n_tiles = (hw + 31) / 32;

  // A fragments: query row q0 + r (zeros beyond Q), all of K in registers
  float4 af[kChunks];
  {
    const bool live = q0 + r < Q;
    const float* a = me + ((long long)b * Q + (live ? q0 + r : 0)) * kC + g * 4;
#pragma unroll
    for (int c = 0; c < kChunks; ++c) af[c] = live ? *reinterpret_cast<const float4*>(a + c * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  unsigned open_acc = 0;  // lanes 0..31: query r of this block, over this wave's tiles

  auto cell_ptr = [&](int t) {
    const int cell = min(t * 32 + r, hw - 1);  // padding cells of the last tile read a real row; they are forced to "blocked"
    return mfd + ((long long)b * hw + cell) * kC + g * 4;
  };
  float4 bf[2][kChunks / 2];
  auto load_half = [&](int t, int half, float4 (&dst)[kChunks / 2]) {
    const float* p = cell_ptr(t) + half * (kChunks / 2) * 8;
#pragma unroll
    for (int c = 0; c < kChunks / 2; ++c) dst[c] = *reinterpret_cast<const float4*>(p + c * 8);
  };
  if (wave < n_tiles) load_half(wave, 0, bf[0]);
  for (int t = wave; t < n_tiles; t += 4) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    load_half(t, 1, bf[1]);  // in flight during the first half's MFMAs
#pragma unroll
    for (int c = 0; c < kChunks / 2; ++c) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c].x, bf[0][c].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c].y, bf[0][c].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c].z, bf[0][c].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c].w, bf[0][c].w, acc, 0, 0, 0);
    }
    if (t + 4 < n_tiles) load_half(t + 4, 0, bf[0]);  // the next tile's first half: in flight during the second half's MFMAs
#pragma unroll
    for (int c = 0; c < kChunks / 2; ++c) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kChunks / 2 + c].x, bf[1][c].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kChunks / 2 + c].y, bf[1][c].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kChunks / 2 + c].z, bf[1][c].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kChunks / 2 + c].w, bf[1][c].w, acc, 0, 0, 0);
    }
    // result layout: acc[i] = S[query (i / 4) * 8 + g * 4 + (i % 4)][cell t * 32 + r]
    const bool real = t * 32 + r < hw;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      // the reference's own test, evaluated the same way (sigmoid(v) < 0.5 differs from v < 0 for |v| below fp32 round-off of 1)
      const bool blk = !real || (1.f / (1.f + expf(-acc[i]))) < 0.5f;
      const unsigned long long ball = __ballot(blk);
      const unsigned long long open = __ballot(!blk);  // (padding cells count as blocked: never "open")
      const int qa = (i >> 2) * 8 + (i & 3);           // query of the g = 0 half; + 4 for g = 1
      if (lane == 0) { words[qa][t] = (unsigned)ball; words[qa + 4][t] = (unsigned)(ball >> 32); }
      // lanes qa / qa + 4 of this wave remember whether their query has any open cell
      if (lane == qa) open_acc |= (unsigned)open;
      if (lane == qa + 4) open_acc |= (unsigned)(open >> 32);
    }
  }
  if (lane < 32) anyopen[wave][lane] = open_acc;
  __syncthreads();
  // ---- row reset (:458) + stores: a fully blocked query row is un-blocked on its real cells --------------------------------------
  for (int i = threadIdx.x; i < 32 * wpitch; i += 256) {
    const int q = i / wpitch, j = i - q * wpitch;
    if (q0 + q >= Q) continue;
    unsigned w = 0xffffffffu;  // words beyond the key tiles: blocked
    if (j < n_tiles) {
      w = words[q][j];
      const bool full = reset_full_rows && !(anyopen[0][q] | anyopen[1][q] | anyopen[2][q] | anyopen[3][q]);
      if (full) {
        const int realbits = min(32, hw - j * 32);
        w = realbits >= 32 ? 0u : (0xffffffffu << realbits);
      }
    }
    bits[((long long)b * Q + q0 + q) * wpitch + j] = w;
  }
  if (bytes) {  // optional byte rows [B, Q, pitch] (1 = blocked; padding blocked) for consumers that want them
    for (int i = threadIdx.x; i < 32 * pitch; i += 256) {
      const int q = i / pitch, c = i - q * pitch;
      if (q0 + q >= Q) continue;
      unsigned char v = 1;
      if (c < hw) {
        const bool full = reset_full_rows && !(anyopen[0][q] | anyopen[1][q] | anyopen[2][q] | anyopen[3][q]);
        v = full ? 0 : (unsigned char)((words[q][c >> 5] >> (c & 31)) & 1u);
      }
      bytes[((long long)b * Q + q0 + q) * pitch + c] = v;
    }
  }
}

}  // namespace

extern "C" {

/* x [B, H*W, C] tokens (a channels-last [B,C,H,W] map) -> out [B, h*w, C], bilinear, align_corners = False:
 * F.interpolate(mask_features, size=(h, w), mode="bilinear") of transformer_decoder.py:502 applied to the PIXEL EMBEDDING
 * instead of to every head's logits (see the file comment).  C % 4 == 0, 16-byte aligned. */
int combo_downsample_tokens_f32(const float* x, int B, int H, int W, int h, int w, int C, float* out, combo_stream_t stream) {
  if (!x || !out || B <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0 || C <= 0 || (C & 3) || (((uintptr_t)x | (uintptr_t)out) & 15))
    return COMBO_EINVAL;
  const long long n = (long long)B * h * w * (C / 4);
  long long grid = (n + 255) / 256;
  if (grid > 256LL * 32) grid = 256LL * 32;
  hipLaunchKernelGGL(downsample_tokens_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, B, H, W, h, w, C, out);
  return (int)hipGetLastError();
}

/* Attention mask of one decoder layer, fused: bits[b, q, j] bit k = (sigmoid(<mask_embed[b, q], mfd[b, 32 j + k]>) < 0.5), cells
 * >= hw blocked, fully blocked rows un-blocked (reset_full_rows; transformer_decoder.py:458, :502-507).  mask_embed [B, Q, 256],
 * mfd [B, hw, 256] (combo_downsample_tokens_f32 of the mask features), bits [B, Q, wpitch] (wpitch >= ceil(hw / 32)), bytes
 * (nullable) [B, Q, pitch] one byte per cell (pitch >= hw).  Exact fp32 (v_mfma_f32_32x32x2_f32). */
int combo_mask_bits_f32(const float* mask_embed, const float* mfd, int B, int Q, int hw, int C, int reset_full_rows, int wpitch,
                        unsigned* bits, int pitch, unsigned char* bytes, combo_stream_t stream) {
  if (!mask_embed || !mfd || !bits || B <= 0 || Q <= 0 || hw <= 0 || C != kC || hw > 32 * kMaxTiles || wpitch < (hw + 31) / 32 ||
      (bytes && pitch < hw) || (((uintptr_t)mask_embed | (uintptr_t)mfd) & 15) || ((uintptr_t)bits & 3))
    return COMBO_EINVAL;
  const int qb = (Q + 31) / 32;
  hipLaunchKernelGGL(mask_bits_kernel, dim3((unsigned)(B * qb)), dim3(256), 0, (hipStream_t)stream, mask_embed, mfd, B, Q, hw,
                     reset_full_rows, wpitch, bits, pitch, bytes);
  return (int)hipGetLastError();
}

}  // extern "C"
