// Hungarian-matcher cost matrices, fused (reference: models/modeling/matcher.py:84-131, batch_sigmoid_ce_loss
// :34-52, batch_dice_loss :13-28, detectron2 point_sample).
//
// For every problem n = (decoder output, frame) and query q the reference point-samples the 56x56 mask logits at
// P = 12 544 shared random points, materialises [Q,P] tensors of softplus(+-x) and sigmoid(x) and contracts them
// with the sampled ground-truth masks ([G,P]) in three GEMMs, once per frame per decoder output (80 problems/step
// at bs = 8, a [80,100,12544] fp32 intermediate = 401 MB when batched).  Here one wave owns one (n, q): its mask
// lives in LDS (12.5 KB), lanes stride over the points, sample bilinearly (zero padding, align_corners=False),
// evaluate softplus / sigmoid in registers and accumulate the 3*G+1 sums the cost needs; nothing of size Q x P
// ever reaches HBM.  The sampled ground truth t[n,g,p] is produced once per n by `gt_sample_kernel`.
//   cost[n,q,g] = w_mask * (sum_p sp(-x) t_g + sp(x) (1 - t_g)) / P  +  w_class * (-softmax(logits[n,q])[label_g])
//               + w_dice * (1 - (2 sum_p s t_g + 1) / (sum_p s + sum_p t_g + 1)),   s = sigmoid(x)
#include "combo_common.h"

namespace {

constexpr int GMAX = 8;     // ground-truth instances per frame supported by the fused kernel
constexpr int WAVES = 8;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
  return v;
}

// bilinear sample with zero padding, align_corners = False (grid_sample semantics): x,y in [0,1] normalised
__device__ __forceinline__ float bilinear(const float* __restrict__ img, int H, int W, float x, float y) {
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

// t[n,g,p] = point_sample(gt[n,g], points[n,p])
__global__ void __launch_bounds__(256)
gt_sample_kernel(const float* __restrict__ gt, const float* __restrict__ points, int N, int G, int H, int W, int P,
                 float* __restrict__ t) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)N * G * P) return;
  const int p = (int)(i % P);
  const long long ng = i / P;
  const int n = (int)(ng / G);
  const float2 xy = *reinterpret_cast<const float2*>(points + ((long long)n * P + p) * 2);
  t[i] = bilinear(gt + ng * H * W, H, W, xy.x, xy.y);
}

// One workgroup (8 waves) per (n, q): the mask sits once in LDS (12.5 KB -> 4 workgroups = 32 waves per CU) and the
// 8 waves split the P points; partial sums meet in a small LDS array.
__global__ void __launch_bounds__(WAVES * 64)
matcher_cost_kernel(const float* __restrict__ logits, const float* __restrict__ masks, const long long* __restrict__ mask_base,
                    const long long* __restrict__ labels,
                    const float* __restrict__ t, const float* __restrict__ points, int N, int Q, int K1, int G, int h, int w,
                    int P, float w_class, float w_mask, float w_dice, float* __restrict__ cost) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float part[WAVES][3 * GMAX + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x / Q, q = blockIdx.x % Q;
  float* img = smem;
  // mask_base[n]: index of query 0's map of problem n inside a larger stack of maps (e.g. the [heads, BT, Q] logits
  // buffer of the decoder: no gathered copy of the ground-truth frames), NULL: maps are packed [N, Q]
  const float* src = masks + ((mask_base ? mask_base[n] : (long long)n * Q) + q) * h * w;
  for (int i = tid; i < h * w; i += WAVES * 64) img[i] = src[i];
  __syncthreads();
  float a[GMAX], d[GMAX], tt[GMAX], s_sum = 0.f;
#pragma unroll
  for (int g = 0; g < GMAX; ++g) { a[g] = 0.f; d[g] = 0.f; tt[g] = 0.f; }
  const float* pts = points + (long long)n * P * 2;
  const float* tn = t + (long long)n * G * P;
  for (int p = tid; p < P; p += WAVES * 64) {
    const float2 xy = *reinterpret_cast<const float2*>(pts + p * 2);
    const float x = bilinear(img, h, w, xy.x, xy.y);
    // F.softplus (beta 1, threshold 20) of +x and -x and sigmoid(x) from ONE exp, ONE log and ONE rcp:
    //   e = exp(-|x|);  softplus(+-x) = max(+-x, 0) + log(1 + e);  sigmoid(x) = (x >= 0 ? 1 : e) / (1 + e)
    const float e = __expf(-fabsf(x));
    const float l1p = __logf(1.f + e);
    float sp_pos = fmaxf(x, 0.f) + l1p;   // softplus(x)
    float sp_neg = fmaxf(-x, 0.f) + l1p;  // softplus(-x)
    if (x > 20.f) sp_pos = x;
    if (x < -20.f) sp_neg = -x;
    const float s = (x >= 0.f ? 1.f : e) * __frcp_rn(1.f + e);
    s_sum += s;
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
      if (g < G) {
        const float tg = tn[g * P + p];
        a[g] += sp_neg * tg + sp_pos * (1.f - tg);
        d[g] += s * tg;
        tt[g] += tg;
      }
    }
  }
  s_sum = wave_sum(s_sum);
  if (lane == 0) part[wave][3 * GMAX] = s_sum;
#pragma unroll
  for (int g = 0; g < GMAX; ++g) {
    if (g < G) {
      const float A = wave_sum(a[g]), D = wave_sum(d[g]), T = wave_sum(tt[g]);
      if (lane == 0) { part[wave][g] = A; part[wave][GMAX + g] = D; part[wave][2 * GMAX + g] = T; }
    }
  }
  __syncthreads();
  if (tid < G) {
    const int g = tid;
    float A = 0.f, D = 0.f, T = 0.f, S = 0.f;
    for (int wv = 0; wv < WAVES; ++wv) {
      A += part[wv][g]; D += part[wv][GMAX + g]; T += part[wv][2 * GMAX + g]; S += part[wv][3 * GMAX];
    }
    const float* lg = logits + ((long long)n * Q + q) * K1;
    float mx = -3.0e38f;
    for (int k = 0; k < K1; ++k) mx = fmaxf(mx, lg[k]);
    float z = 0.f;
    for (int k = 0; k < K1; ++k) z += expf(lg[k] - mx);
    const int lab = (int)labels[(long long)n * G + g];
    const float prob = expf(lg[lab] - mx) / z;
    cost[((long long)n * Q + q) * G + g] =
        w_mask * (A / (float)P) + w_class * (-prob) + w_dice * (1.f - (2.f * D + 1.f) / (S + T + 1.f));
  }
}

}  // namespace

extern "C" int combo_matcher_cost_f32(const float* logits, const float* masks, const long long* mask_base,
                                      const long long* labels, const float* gt,
                                      const float* points, int N, int Q, int K1, int G, int h, int w, int H, int W,
                                      int P, float w_class, float w_mask, float w_dice, float* t_ws, float* cost,
                                      combo_stream_t stream) {
  if (!logits || !masks || !labels || !gt || !points || !t_ws || !cost || N <= 0 || Q <= 0 || K1 <= 0 || G <= 0 ||
      G > GMAX || h <= 0 || w <= 0 || P <= 0 || (size_t)h * w * 4 > 150 * 1024)
    return COMBO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long long tot = (long long)N * G * P;
  hipLaunchKernelGGL(gt_sample_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, gt, points, N, G, H, W, P, t_ws);
  const size_t lds = (size_t)h * w * 4;
  static ComboDevFlag attr;
  if (lds > 64 * 1024 && !attr.is_set()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(matcher_cost_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr.mark();
  }
  hipLaunchKernelGGL(matcher_cost_kernel, dim3(N * Q), dim3(WAVES * 64), lds, st, logits, masks, mask_base, labels, t_ws,
                     points, N, Q, K1, G, h, w, P, w_class, w_mask, w_dice, cost);
  return (int)hipGetLastError();
}
