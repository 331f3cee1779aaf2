// Mask losses of the criterion, fused (reference: models/modeling/criterion.py:137-186 loss_masks, :19-62 dice /
// sigmoid-CE, :70-84 calculate_uncertainty; detectron2 get_uncertain_point_coords_with_randomness + point_sample).
//
// For every matched (prediction, target) pair n (2 per GT frame x 8 frames x 10 decoder outputs = 160 per step) the
// reference (a) point-samples the 56x56 logit map at 3x12544 random points, takes the 9408 most uncertain ones
// (top-k of -|logit|: a device sort per mask) and appends 3136 fresh random points, (b) point-samples prediction and
// 224x224 target at those 12544 points, (c) evaluates BCE-with-logits (mean over points) and dice; backward scatters
// through grid_sample's atomics.  ~60 small launches per decoder output.  Here: three launches for all pairs.
//   select : one workgroup per pair; the logit map sits in LDS; the k-th smallest |logit| is found by a 3-pass radix
//            select on the float bits (11+11+9 bits, integer LDS histograms), then the selected points are compacted
//            in index order (deterministic, ties resolved by index) - no sort.
//   fwd    : one workgroup per pair: sample prediction (LDS) and target (L2), accumulate sum BCE, sum s*t, sum s, sum t.
//   bwd    : recompute the samples, scatter d loss / d logit through the 4 bilinear taps into an LDS gradient map
//            (ds_add_f32: 50 k lane-ops per pair, one pair per CU, so the slow LDS float atomic is affordable here),
//            plain stores to the dense gradient tensor (matched (output, frame, query) triples are unique).
// Pairs address their maps through flat indices into the stacked prediction tensor [L*F*Q, h, w] and the padded
// target tensor [F*Gmax, H, W]: no gather copies.
#include "combo_common.h"

namespace {

constexpr int THREADS = 512;
constexpr int NWAVE = THREADS / 64;

__device__ __forceinline__ float bilinear_lds(const float* __restrict__ img, int H, int W, float x, float y) {
  const float fx = x * W - 0.5f, fy = y * H - 0.5f;
  const float x0f = floorf(fx), y0f = floorf(fy);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const float lx = fx - x0f, ly = fy - y0f;
  const bool xl = x0 >= 0 && x0 < W, xr = x0 + 1 >= 0 && x0 + 1 < W, yt = y0 >= 0 && y0 < H, yb = y0 + 1 >= 0 && y0 + 1 < H;
  const float v00 = (xl && yt) ? img[y0 * W + x0] : 0.f;
  const float v01 = (xr && yt) ? img[y0 * W + x0 + 1] : 0.f;
  const float v10 = (xl && yb) ? img[(y0 + 1) * W + x0] : 0.f;
  const float v11 = (xr && yb) ? img[(y0 + 1) * W + x0 + 1] : 0.f;
  return (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NWAVE; ++w) t += red[w];
  return t;
}

// Smallest bin b with cumulative count >= k.  hist[nbins] in LDS; returns b and (k - count of bins < b) through
// shared outputs.  Executed by wave 0.
__device__ __forceinline__ void find_bin(const int* __restrict__ hist, int nbins, int k, int* __restrict__ out_bin,
                                         int* __restrict__ out_rem) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const int per = (nbins + 63) / 64;
    int local = 0;
    for (int j = 0; j < per; ++j) {
      const int b = lane * per + j;
      if (b < nbins) local += hist[b];
    }
    int incl = local;  // inclusive scan over lanes
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const int o = __shfl_up(incl, s);
      if (lane >= s) incl += o;
    }
    const int excl = incl - local;
    if (excl < k && k <= incl) {  // the k-th element lives in this lane's bins
      int c = excl;
      for (int j = 0; j < per; ++j) {
        const int b = lane * per + j;
        const int hb = b < nbins ? hist[b] : 0;
        if (c + hb >= k) { *out_bin = b; *out_rem = k - c; break; }
        c += hb;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// select: coords_out[n, 0:k] = the k points of over[n] with the smallest |logit|; coords_out[n, k:] = extra[n]
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(THREADS)
uncertain_select_kernel(const float* __restrict__ masks, const long long* __restrict__ midx, int h, int w,
                        const float* __restrict__ over, int NS, const float* __restrict__ extra, int NR, int k,
                        float* __restrict__ coords_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* img = smem;                                   // h*w
  int* hist = reinterpret_cast<int*>(smem + h * w);   // 2048
  __shared__ int s_bin, s_rem, wave_lt[NWAVE], wave_eq[NWAVE], base_lt, base_eq;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x;
  const float* src = masks + midx[n] * h * w;
  for (int i = tid; i < h * w; i += THREADS) img[i] = src[i];
  const float* pts = over + (long long)n * NS * 2;
  unsigned prefix = 0;  // bits fixed so far, and their mask
  unsigned pmask = 0;
  int kk = k;
  const int shifts[3] = {20, 9, 0};
  const int widths[3] = {11, 11, 9};
  for (int pass = 0; pass < 3; ++pass) {
    const int nb = 1 << widths[pass];
    for (int i = tid; i < nb; i += THREADS) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < NS; i += THREADS) {
      const float2 xy = *reinterpret_cast<const float2*>(pts + i * 2);
      const unsigned key = __float_as_uint(fabsf(bilinear_lds(img, h, w, xy.x, xy.y)));
      if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shifts[pass]) & (nb - 1)], 1);
    }
    __syncthreads();
    find_bin(hist, nb, kk, &s_bin, &s_rem);
    __syncthreads();
    prefix |= ((unsigned)s_bin) << shifts[pass];
    pmask |= ((unsigned)(nb - 1)) << shifts[pass];
    kk = s_rem;
    __syncthreads();
  }
  const unsigned T = prefix;  // the k-th smallest key; kk = how many keys == T to take
  if (tid == 0) { base_lt = 0; base_eq = 0; }
  __syncthreads();
  float* out = coords_out + (long long)n * (k + NR) * 2;
  for (int i0 = 0; i0 < NS; i0 += THREADS) {
    const int i = i0 + tid;
    float2 xy = make_float2(0.f, 0.f);
    bool lt = false, eq = false;
    if (i < NS) {
      xy = *reinterpret_cast<const float2*>(pts + i * 2);
      const unsigned key = __float_as_uint(fabsf(bilinear_lds(img, h, w, xy.x, xy.y)));
      lt = key < T;
      eq = key == T;
    }
    const unsigned long long bl = __ballot(lt), be = __ballot(eq);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) { wave_lt[wave] = __popcll(bl); wave_eq[wave] = __popcll(be); }
    __syncthreads();
    int lt_before = base_lt + __popcll(bl & below), eq_before = base_eq + __popcll(be & below);
    for (int wv = 0; wv < wave; ++wv) { lt_before += wave_lt[wv]; eq_before += wave_eq[wv]; }
    if (lt || (eq && eq_before < kk)) {
      const int pos = lt_before + min(eq_before, kk);
      *reinterpret_cast<float2*>(out + pos * 2) = xy;
    }
    __syncthreads();
    if (tid == 0) {
      int a = 0, b = 0;
      for (int wv = 0; wv < NWAVE; ++wv) { a += wave_lt[wv]; b += wave_eq[wv]; }
      base_lt += a;
      base_eq += b;
    }
    __syncthreads();
  }
  const float* ex = extra + (long long)n * NR * 2;
  for (int i = tid; i < NR; i += THREADS)
    *reinterpret_cast<float2*>(out + (k + i) * 2) = *reinterpret_cast<const float2*>(ex + i * 2);
}

// global-memory bilinear (targets, 224x224)
__device__ __forceinline__ float bilinear_glb(const float* __restrict__ img, int H, int W, float x, float y) {
  return bilinear_lds(img, H, W, x, y);
}

// ---------------------------------------------------------------------------------------------------------------
// fwd: stats[n] = (sum_p BCE(x_p, t_p), sum_p s_p t_p, sum_p s_p, sum_p t_p)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(THREADS)
mask_loss_fwd_kernel(const float* __restrict__ masks, const long long* __restrict__ midx, int h, int w,
                     const float* __restrict__ gt, const long long* __restrict__ gidx, int H, int W,
                     const float* __restrict__ coords, int P, float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[NWAVE];
  float* img = smem;
  const int tid = threadIdx.x, n = blockIdx.x;
  const float* src = masks + midx[n] * h * w;
  for (int i = tid; i < h * w; i += THREADS) img[i] = src[i];
  __syncthreads();
  const float* tgt = gt + gidx[n] * (long long)H * W;
  const float* pts = coords + (long long)n * P * 2;
  float bce = 0.f, A = 0.f, Bs = 0.f, C = 0.f;
  for (int i = tid; i < P; i += THREADS) {
    const float2 xy = *reinterpret_cast<const float2*>(pts + i * 2);
    const float x = bilinear_lds(img, h, w, xy.x, xy.y);
    const float t = bilinear_glb(tgt, H, W, xy.x, xy.y);
    const float e = __expf(-fabsf(x));
    bce += fmaxf(x, 0.f) - x * t + log1pf(e);  // binary_cross_entropy_with_logits
    const float s = (x >= 0.f ? 1.f : e) / (1.f + e);
    A += s * t;
    Bs += s;
    C += t;
  }
  const float r0 = block_sum(bce, red), r1 = block_sum(A, red), r2 = block_sum(Bs, red), r3 = block_sum(C, red);
  if (tid == 0) {
    stats[n * 4 + 0] = r0; stats[n * 4 + 1] = r1; stats[n * 4 + 2] = r2; stats[n * 4 + 3] = r3;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// bwd: grad_masks[midx[n]] = d/d logits of ( g_bce[n] * mean_p BCE + g_dice[n] * (1 - (2A+1)/(B+C+1)) )
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(THREADS)
mask_loss_bwd_kernel(const float* __restrict__ masks, const long long* __restrict__ midx, int h, int w,
                     const float* __restrict__ gt, const long long* __restrict__ gidx, int H, int W,
                     const float* __restrict__ coords, int P, const float* __restrict__ stats,
                     const float* __restrict__ g_bce, const float* __restrict__ g_dice, float* __restrict__ grad_masks,
                     int accumulate, const long long* __restrict__ grad_index) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* img = smem;
  float* gimg = smem + h * w;
  const int tid = threadIdx.x, n = blockIdx.x;
  const float* src = masks + midx[n] * h * w;
  for (int i = tid; i < h * w; i += THREADS) { img[i] = src[i]; gimg[i] = 0.f; }
  __syncthreads();
  const float* tgt = gt + gidx[n] * (long long)H * W;
  const float* pts = coords + (long long)n * P * 2;
  const float A = stats[n * 4 + 1], Bs = stats[n * 4 + 2], C = stats[n * 4 + 3];
  const float den = Bs + C + 1.f, num = 2.f * A + 1.f;
  const float gb = g_bce[n] / (float)P, gd = g_dice[n];
  for (int i = tid; i < P; i += THREADS) {
    const float2 xy = *reinterpret_cast<const float2*>(pts + i * 2);
    const float fx = xy.x * w - 0.5f, fy = xy.y * h - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float lx = fx - x0f, ly = fy - y0f;
    const bool xl = x0 >= 0 && x0 < w, xr = x0 + 1 >= 0 && x0 + 1 < w, yt = y0 >= 0 && y0 < h, yb = y0 + 1 >= 0 && y0 + 1 < h;
    const float v00 = (xl && yt) ? img[y0 * w + x0] : 0.f, v01 = (xr && yt) ? img[y0 * w + x0 + 1] : 0.f;
    const float v10 = (xl && yb) ? img[(y0 + 1) * w + x0] : 0.f, v11 = (xr && yb) ? img[(y0 + 1) * w + x0 + 1] : 0.f;
    const float x = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    const float t = bilinear_glb(tgt, H, W, xy.x, xy.y);
    const float e = __expf(-fabsf(x));
    const float s = (x >= 0.f ? 1.f : e) / (1.f + e);
    // d BCE / dx = s - t ;  d dice / dx = -(2 t den - num) / den^2 * s (1 - s)
    const float dx = gb * (s - t) - gd * (2.f * t * den - num) / (den * den) * s * (1.f - s);
    if (xl && yt) atomicAdd(&gimg[y0 * w + x0], dx * (1.f - ly) * (1.f - lx));
    if (xr && yt) atomicAdd(&gimg[y0 * w + x0 + 1], dx * (1.f - ly) * lx);
    if (xl && yb) atomicAdd(&gimg[(y0 + 1) * w + x0], dx * ly * (1.f - lx));
    if (xr && yb) atomicAdd(&gimg[(y0 + 1) * w + x0 + 1], dx * ly * lx);
  }
  __syncthreads();
  float* dst = grad_masks + (grad_index ? grad_index[n] : midx[n]) * h * w;
  // accumulate: the map already holds another term's gradient (cosine loss written by combo_cosine_grad_f32); a map is
  // matched at most once per frame and output, so the read-modify-write needs no atomics
  if (accumulate) { for (int i = tid; i < h * w; i += THREADS) dst[i] += gimg[i]; }
  else { for (int i = tid; i < h * w; i += THREADS) dst[i] = gimg[i]; }
}

// the mask maps of a 512 x 512 input are 128 x 128 fp32 = 64 KiB (+ as much again for the gradient map): above the 64 KiB a
// kernel gets without asking
constexpr size_t kMaxLds = 152 * 1024;
int big_lds(const void* fn) {
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

extern "C" {

int combo_uncertain_points_f32(const float* masks, const long long* mask_index, int NM, int h, int w, const float* over_points,
                               int NS, const float* extra_points, int NR, int k, float* coords_out, combo_stream_t stream) {
  if (!masks || !mask_index || !over_points || !coords_out || NM <= 0 || h <= 0 || w <= 0 || NS <= 0 || k <= 0 || k > NS ||
      NR < 0 || (NR > 0 && !extra_points) || (size_t)h * w * 4 + 2048 * 4 > kMaxLds)
    return COMBO_EINVAL;
  if (int e = big_lds(reinterpret_cast<const void*>(uncertain_select_kernel))) return e;
  hipLaunchKernelGGL(uncertain_select_kernel, dim3(NM), dim3(THREADS), (size_t)h * w * 4 + 2048 * 4, (hipStream_t)stream, masks,
                     mask_index, h, w, over_points, NS, extra_points, NR, k, coords_out);
  return (int)hipGetLastError();
}

int combo_mask_loss_forward_f32(const float* masks, const long long* mask_index, int NM, int h, int w, const float* gt,
                                const long long* gt_index, int H, int W, const float* coords, int P, float* stats,
                                combo_stream_t stream) {
  if (!masks || !mask_index || !gt || !gt_index || !coords || !stats || NM <= 0 || P <= 0 || (size_t)h * w * 4 > kMaxLds)
    return COMBO_EINVAL;
  if (int e = big_lds(reinterpret_cast<const void*>(mask_loss_fwd_kernel))) return e;
  hipLaunchKernelGGL(mask_loss_fwd_kernel, dim3(NM), dim3(THREADS), (size_t)h * w * 4, (hipStream_t)stream, masks, mask_index,
                     h, w, gt, gt_index, H, W, coords, P, stats);
  return (int)hipGetLastError();
}

int combo_mask_loss_backward_f32(const float* masks, const long long* mask_index, int NM, int h, int w, const float* gt,
                                 const long long* gt_index, int H, int W, const float* coords, int P, const float* stats,
                                 const float* g_bce, const float* g_dice, float* grad_masks, int accumulate,
                                 const long long* grad_index, combo_stream_t stream) {
  if (!masks || !mask_index || !gt || !gt_index || !coords || !stats || !g_bce || !g_dice || !grad_masks || NM <= 0 ||
      P <= 0 || (size_t)h * w * 8 > kMaxLds)
    return COMBO_EINVAL;
  if (int e = big_lds(reinterpret_cast<const void*>(mask_loss_bwd_kernel))) return e;
  hipLaunchKernelGGL(mask_loss_bwd_kernel, dim3(NM), dim3(THREADS), (size_t)h * w * 8, (hipStream_t)stream, masks, mask_index,
                     h, w, gt, gt_index, H, W, coords, P, stats, g_bce, g_dice, grad_masks, accumulate, grad_index);
  return (int)hipGetLastError();
}

}  // extern "C"
