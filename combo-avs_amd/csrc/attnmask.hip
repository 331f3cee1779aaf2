// Next-layer attention mask from mask logits (reference: transformer_decoder.py:502-507 + the row reset at :458).
//
//   am = F.interpolate(logits [N,H,W], size=(h,w), mode="bilinear", align_corners=False)
//   blocked = sigmoid(am) < 0.5 ;  rows whose h*w cells are ALL blocked are fully unblocked (:458)
//
// The reference materialises the interpolated map, the sigmoid and an 8x head-replicated bool tensor and needs a
// nonzero()+index_put (host sync) for the row reset.  Here: one wave per (frame, query) row reads the <= 4*h*w
// source taps it needs (for the 56->7/14/28 cases: the central 2x2 of every s x s cell), thresholds, decides the
// row reset with a wave ballot and writes h*w bytes.  Bilinear arithmetic follows ATen's upsample_bilinear2d
// (area_pixel_compute_source_index, align_corners=False; src<0 clamps to 0).
#include "combo_common.h"

namespace {

__global__ void __launch_bounds__(256)
attn_mask_kernel(const float* __restrict__ logits, int N, int H, int W, int h, int w, int reset_full_rows, int pitch,
                 unsigned char* __restrict__ blocked, int wpitch, unsigned* __restrict__ bitrows) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= N) return;
  const float* src = logits + (long long)wave * H * W;
  unsigned char* dst = blocked + (long long)wave * pitch;  // row pitch >= h*w; the padding cells are written as blocked
  const float sh = (float)H / (float)h, sw = (float)W / (float)w;
  const int n = h * w;
  bool all_blocked = true;
  // up to 64 cells per lane kept in a bitmask (h*w <= 4096: the 64 x 64 level of a 512 x 512 input)
  unsigned long long bits = 0;
  int cnt = 0;
  for (int i = lane; i < n; i += 64, ++cnt) {
    const int oy = i / w, ox = i - oy * w;
    float fy = sh * (oy + 0.5f) - 0.5f;
    float fx = sw * (ox + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int yp = (y0 < H - 1) ? 1 : 0, xp = (x0 < W - 1) ? 1 : 0;
    const float ly1 = fy - y0, ly0 = 1.f - ly1, lx1 = fx - x0, lx0 = 1.f - lx1;
    const float* p = src + y0 * W + x0;
    const float v = ly0 * (lx0 * p[0] + lx1 * p[xp]) + ly1 * (lx0 * p[yp * W] + lx1 * p[yp * W + xp]);
    const bool b = (1.f / (1.f + expf(-v))) < 0.5f;
    bits |= (b ? 1ull : 0ull) << cnt;
    all_blocked &= b;
  }
  const bool row_full = __all(all_blocked) && reset_full_rows;
  if (blocked) {
    cnt = 0;
    for (int i = lane; i < n; i += 64, ++cnt) dst[i] = row_full ? 0 : (unsigned char)((bits >> cnt) & 1ull);
    for (int i = n + lane; i < pitch; i += 64) dst[i] = 1;
  }
  if (bitrows) {
    // the same row bit-packed for the attention kernels: bit k of word j = cell 32 j + k; cells >= h*w read as blocked
    unsigned* brow = bitrows + (long long)wave * wpitch;
    const int chunks = (n + 63) / 64;
    for (int cc = 0; cc < chunks; ++cc) {
      const bool b = (cc * 64 + lane >= n) || (!row_full && ((bits >> cc) & 1ull));
      const unsigned long long word = __ballot(b);
      if (lane == 0) {
        brow[2 * cc] = (unsigned)word;
        if (2 * cc + 1 < wpitch) brow[2 * cc + 1] = (unsigned)(word >> 32);
      }
    }
    for (int j = 2 * chunks + lane; j < wpitch; j += 64) brow[j] = 0xffffffffu;
  }
}

}  // namespace

extern "C" int combo_attn_mask_pitched_f32(const float* logits, int N, int H, int W, int h, int w, int reset_full_rows, int pitch,
                                           unsigned char* blocked, combo_stream_t stream) {
  if (!logits || !blocked || N <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0 || h * w > 4096 || pitch < h * w) return COMBO_EINVAL;
  const int waves_per_block = 4;
  hipLaunchKernelGGL(attn_mask_kernel, dim3((N + waves_per_block - 1) / waves_per_block), dim3(256), 0,
                     (hipStream_t)stream, logits, N, H, W, h, w, reset_full_rows, pitch, blocked, 0, (unsigned*)nullptr);
  return (int)hipGetLastError();
}

extern "C" int combo_attn_mask_bits_f32(const float* logits, int N, int H, int W, int h, int w, int reset_full_rows, int pitch,
                                        unsigned char* blocked, int wpitch, unsigned* bits, combo_stream_t stream) {
  if (!logits || !bits || N <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0 || h * w > 4096 || (blocked && pitch < h * w) ||
      wpitch < (h * w + 31) / 32)
    return COMBO_EINVAL;
  const int waves_per_block = 4;
  hipLaunchKernelGGL(attn_mask_kernel, dim3((N + waves_per_block - 1) / waves_per_block), dim3(256), 0,
                     (hipStream_t)stream, logits, N, H, W, h, w, reset_full_rows, pitch, blocked, wpitch, bits);
  return (int)hipGetLastError();
}

extern "C" int combo_attn_mask_f32(const float* logits, int N, int H, int W, int h, int w, int reset_full_rows,
                                   unsigned char* blocked, combo_stream_t stream) {
  return combo_attn_mask_pitched_f32(logits, N, H, W, h, w, reset_full_rows, h * w, blocked, stream);
}
