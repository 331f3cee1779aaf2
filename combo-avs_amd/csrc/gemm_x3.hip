// Entry points of the 3-product bf16 GEMM family (fp32 accuracy from hi.hi + lo.hi + hi.lo on the bf16 matrix cores) and the weight
// pre-split they consume.  The kernel itself is csrc/gemm_nt3.hip (round 4); round 3's gemm_nt2 kernel - two workgroups per CU
// whose phases did not overlap: 0.18 of its MFMA ceiling inside the training step - is gone.
//   combo_presplit_bf16x2_*          W (any strided [N, K] view, e.g. W^T) -> image: per 8 k a 16-B bf16 `hi` group (hi = rne(x)) and
//                                    a 16-B bf16 `lo` group (lo = rne(x - hi)), 4 bytes per element, row pitch K floats
//   combo_gemm_nt_x3_pre_*           C = A . image^T (+ bias) (+ ReLU) (masked) (batched)
//   combo_conv3x3_nhwc_x3_pre_f32    implicit-GEMM 3x3 convolution over NHWC tokens (the input gradient of a 3x3 convolution with
//                                    flipped taps; the head's bf16 forward mode)
//   combo_gemm_nt2_products          1 = plain bf16 (the head's bf16 throughput mode), 3 = the fp32-accurate split (default)
#include <cstdlib>

#include "combo_common.h"
#include "gemm_nt3.h"

namespace {

__device__ __forceinline__ unsigned pack_rne(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ unsigned pack_rne_f16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
// one packed pair of `hi` pieces and of `lo` pieces of two fp32 values: bf16 (hi = rne(x), lo = rne(x - hi)) or, f16, the same on
// fp16 pieces (csrc/gemm_nt3.hip: the fp32-grade forward mode; |x| < 65 504)
// f16: 0 = bf16 pieces, 1 = fp16 pieces of 2^8 . x (a weight image), 2 = fp16 pieces of x (an activation image)
__device__ __forceinline__ void split_pair(float a, float b, int f16, unsigned& h, unsigned& l) {
  if (f16) {
    if (f16 == 1) {
      a *= (float)(1 << COMBO_F16_BSCALE_LOG2);  // (exact; undone by the GEMM's epilogue, csrc/gemm_nt3.hip)
      b *= (float)(1 << COMBO_F16_BSCALE_LOG2);
    }
    h = pack_rne_f16(a, b);
    const f16x2 v = __builtin_bit_cast(f16x2, h);
    l = pack_rne_f16(a - (float)v[0], b - (float)v[1]);
  } else {
    h = pack_rne(a, b);
    l = pack_rne(a - __uint_as_float(h << 16), b - __uint_as_float(h & 0xffff0000u));
  }
}

constexpr int kBK = 16;

// Weight image: element (n, k) = src[n*ld_row + k*ld_col]; per 8 consecutive k a 16-B group of bf16 `hi`
// = rne(x) followed by a 16-B group of bf16 `lo` = rne(x - hi): 4 bytes per element, row stride K floats.
__global__ void __launch_bounds__(256)
presplit_kernel(const float* __restrict__ src, long long ld_row, long long ld_col, int N, int K, uint4* __restrict__ img,
                long long batch_stride, int f16) {
  const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int kg = K >> 3;
  if (t >= (long long)N * kg) return;
  src += blockIdx.y * batch_stride;               // batch entry b: source at src + b*batch_stride, image at img + b*N*K floats
  img += blockIdx.y * ((long long)N * kg * 2);
  // neighbouring threads walk the unit-stride direction of the source: k groups for W, rows for a W^T view
  int n, g8;
  if (ld_col == 1) { n = (int)(t / kg); g8 = (int)(t - (long long)n * kg); }
  else { g8 = (int)(t / N); n = (int)(t - (long long)g8 * N); }
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = src[(long long)n * ld_row + (long long)(g8 * 8 + i) * ld_col];
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split_pair(v[2 * i], v[2 * i + 1], f16, h[i], l[i]);
  const long long o = ((long long)n * kg + g8) * 2;
  img[o] = make_uint4(h[0], h[1], h[2], h[3]);
  img[o + 1] = make_uint4(l[0], l[1], l[2], l[3]);
}

// Grouped form: every weight whose input-gradient GEMM the backward pass will run, split by ONE launch (per 56 problems)
// instead of one ~5 us launch per weight (148 per training step).  img_ld: floats per image row (>= K): several sources may
// fill k ranges of one image (the two weights of a column-concatenated layer).
constexpr int kMaxSplitGroup = 56;  // (the argument block of a launch stays below 4 KiB)
struct SplitGroupArgs {
  int count, f16;
  long long thread_start[kMaxSplitGroup + 1];
  combo_presplit_problem p[kMaxSplitGroup];
};

__global__ void __launch_bounds__(256)
presplit_grouped_kernel(const SplitGroupArgs args) {
  const long long t0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t0 >= args.thread_start[args.count]) return;
  int pi = 0;
  for (int i = 1; i < args.count; ++i)
    if (t0 >= args.thread_start[i]) pi = i;
  const combo_presplit_problem& pr = args.p[pi];
  const long long t = t0 - args.thread_start[pi];
  const int kg = pr.K >> 3;
  if (t >= (long long)pr.N * kg) return;  // (a problem's threads are rounded up to whole workgroups)
  // neighbouring threads walk the direction in which the source is denser: the k groups of a row or the rows of a k group
  const long long row_step = pr.ld_row < 0 ? -pr.ld_row : pr.ld_row, grp_step = 8 * (pr.ld_col < 0 ? -pr.ld_col : pr.ld_col);
  int n, g8;
  if (grp_step <= row_step) { n = (int)(t / kg); g8 = (int)(t - (long long)n * kg); }
  else { g8 = (int)(t / pr.N); n = (int)(t - (long long)g8 * pr.N); }
  const int taps = pr.taps > 1 ? pr.taps : 1;
  const float* src = pr.src + (long long)n * pr.ld_row + (long long)(g8 * 8) * pr.ld_col;
  for (int tap = 0; tap < taps; ++tap) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = src[(long long)i * pr.ld_col + tap];
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_pair(v[2 * i], v[2 * i + 1], args.f16, h[i], l[i]);
    const int col_tap = pr.flip ? taps - 1 - tap : tap;
    uint4* img = reinterpret_cast<uint4*>(pr.img) + (long long)n * (pr.img_ld >> 2) + ((long long)col_tap * kg + g8) * 2;
    img[0] = make_uint4(h[0], h[1], h[2], h[3]);
    img[1] = make_uint4(l[0], l[1], l[2], l[3]);
  }
}

int g_products = 3;  // combo_gemm_nt2_products: bf16 products per multiply-add of the launches that follow (host state, read at launch)
int g_split_f16 = 0;  // combo_presplit_pieces: the pre-split launches that follow write fp16 pieces (1) or bf16 pieces (0)

}  // namespace

/* bf16 products per fp32 multiply-add of the combo_gemm_nt_x3_* / combo_conv3x3_nhwc_x3_* launches that FOLLOW (3 = the
 * fp32-accurate split, the default; 1 = plain bf16 inputs with fp32 accumulation: the head's bf16 throughput mode, forward GEMMs
 * only).  Returns the previous value.  Host-side state, read when a launch is issued (a captured graph keeps what was set). */
extern "C" int combo_gemm_nt2_products(int products) {
  const int prev = g_products;
  if (products == 1 || products == 3 || products == COMBO_PRODUCTS_F16X3) g_products = products;
  return prev;
}

/* Piece type of the combo_presplit_bf16x2_* launches that FOLLOW: 0 = bf16 hi / lo (the default: every gradient GEMM), 1 = fp16
 * hi / lo (the image of a FORWARD weight for combo_gemm_nt2_products(19): 22 mantissa bits, |w| < 65 504).  Returns the previous
 * value.  Host-side state, read when a launch is issued. */
extern "C" int combo_presplit_pieces(int f16) {
  const int prev = g_split_f16;
  if (f16 == 0 || f16 == 1) g_split_f16 = f16;
  return prev;
}

extern "C" int combo_presplit_bf16x2_f32(const float* src, long long ld_row, long long ld_col, int N, int K, float* img,
                                         combo_stream_t stream) {
  if (!src || !img || N <= 0 || K <= 0 || K % 8 != 0 || ((uintptr_t)img & 15)) return COMBO_EINVAL;
  const long long threads = (long long)N * (K / 8);
  hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld_row,
                     ld_col, N, K, reinterpret_cast<uint4*>(img), 0LL, g_split_f16);
  return (int)hipGetLastError();
}

extern "C" int combo_presplit_bf16x2_grouped_f32(const combo_presplit_problem* problems, int count, combo_stream_t stream) {
  if (!problems || count <= 0) return COMBO_EINVAL;
  for (int base = 0; base < count; base += kMaxSplitGroup) {
    SplitGroupArgs a;
    a.count = count - base < kMaxSplitGroup ? count - base : kMaxSplitGroup;
    a.f16 = g_split_f16;
    long long threads = 0;
    for (int i = 0; i < a.count; ++i) {
      const combo_presplit_problem& pr = problems[base + i];
      if (!pr.src || !pr.img || pr.N <= 0 || pr.K <= 0 || pr.K % 8 != 0 || pr.taps < 0 || pr.taps > 9 ||
          pr.img_ld < (long long)pr.K * (pr.taps > 1 ? pr.taps : 1) || pr.img_ld % 8 != 0 || ((uintptr_t)pr.img & 31))
        return COMBO_EINVAL;
      a.thread_start[i] = threads;
      a.p[i] = pr;
      threads += ((long long)pr.N * (pr.K / 8) + 255) / 256 * 256;  // a workgroup never straddles two problems
    }
    a.thread_start[a.count] = threads;
    hipLaunchKernelGGL(presplit_grouped_kernel, dim3((unsigned)(threads / 256)), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

extern "C" int combo_presplit_bf16x2_batched_f32(const float* src, long long ld_row, long long ld_col, long long batch_stride,
                                                 int N, int K, int batch, float* img, combo_stream_t stream) {
  if (!src || !img || N <= 0 || K <= 0 || K % 8 != 0 || batch <= 0 || batch > 65535 || ((uintptr_t)img & 15)) return COMBO_EINVAL;
  const long long threads = (long long)N * (K / 8);
  hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)((threads + 255) / 256), (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
                     src, ld_row, ld_col, N, K, reinterpret_cast<uint4*>(img), batch_stride, g_split_f16 ? 2 : 0);  // (activations: unscaled)
  return (int)hipGetLastError();
}

// `batch` independent GEMMs of one shape (the mask-logit contraction mask_embed @ pixel_embed^T per frame and its input
// gradient): C_b[M,N] = A_b[M,K] . B_b[N,K]^T, operands at base + b * stride (elements).  128 x 128 tiles when M pads better to
// 128 than to 256.
extern "C" int combo_gemm_nt_x3_pre_batched_f32(const float* A, long long lda, long long sA, const float* Bimg, long long sB,
                                                float* C, long long ldc, long long sC, int M, int N, int K, int batch, int relu,
                                                combo_stream_t stream) {
  if (!A || !Bimg || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || K % kBK != 0 || lda % 4 != 0 || sA % 4 != 0 ||
      sB % 4 != 0 || ((uintptr_t)A & 15) || ((uintptr_t)Bimg & 15))
    return COMBO_EINVAL;
  const int pad256 = (M + 255) / 256 * 256, pad128 = (M + 127) / 128 * 128;
  // M = 100 queries per frame pads to 128, not 256: 128 x 128 tiles; otherwise the planner weighs tile rounds (M = 1000 rows of 10
  // heads x 40 frames: 320 wide tiles are 2 rounds on 256 CUs, 640 mid tiles 3 cheaper ones)
  // (fp16 pieces: the batched form's images are ACTIVATIONS - combo_presplit_bf16x2_batched_f32 splits them unscaled)
  return combo_nt3_launch(A, lda, Bimg, K, nullptr, nullptr, C, ldc, M, N, K, relu,
                          g_products == COMBO_PRODUCTS_F16X3 ? COMBO_PRODUCTS_F16X3_UNSCALED : g_products, batch, sA, sB, sC, nullptr,
                          pad128 < pad256 ? 2 : 0, stream);
}

extern "C" int combo_gemm_nt_x3_pre_f32(const float* A, long long lda, const float* Bimg, const float* bias, float* C,
                                        long long ldc, int M, int N, int K, int relu, combo_stream_t stream) {
  if (!A || !Bimg || !C || M <= 0 || N <= 0 || K <= 0 || K % kBK != 0 || lda % 4 != 0 || ((uintptr_t)A & 15) ||
      ((uintptr_t)Bimg & 15))
    return COMBO_EINVAL;
  return combo_nt3_launch(A, lda, Bimg, K, bias, nullptr, C, ldc, M, N, K, relu, g_products, 1, 0, 0, 0, nullptr, 0, stream);
}

/* combo_gemm_nt_x3_pre_f32 with split-K (splits = combo_gemm_nt_x3_splitk_plan(M, N, K) > 1; workspace [splits, M, N]): the forward
 * GEMMs with a long reduction and few output tiles (the decoder FFN's linear2 4000 x 2048 -> 256, the res5 / res4 input projections)
 * in the 3-product forward modes - what combo_gemm_nt_splitk_f32 is to the exact path. */
extern "C" int combo_gemm_nt_x3_pre_splitk_f32(const float* A, long long lda, const float* Bimg, const float* bias, float* C,
                                               long long ldc, int M, int N, int K, int relu, int splits, float* workspace,
                                               combo_stream_t stream) {
  if (!A || !Bimg || !C || M <= 0 || N <= 0 || K <= 0 || K % kBK != 0 || lda % 4 != 0 || ((uintptr_t)A & 15) || ((uintptr_t)Bimg & 15))
    return COMBO_EINVAL;
  return nt3_split_launch(A, lda, Bimg, bias, nullptr, nullptr, C, ldc, M, N, K, relu, splits, workspace, nullptr, stream, g_products);
}

extern "C" int combo_gemm_nt_x3_pre_masked_f32(const float* A, long long lda, const float* Bimg, const float* mask, float* C,
                                               long long ldc, int M, int N, int K, combo_stream_t stream) {
  if (!A || !Bimg || !C || !mask || M <= 0 || N <= 0 || K <= 0 || K % kBK != 0 || lda % 4 != 0 || ((uintptr_t)A & 15) ||
      ((uintptr_t)Bimg & 15))
    return COMBO_EINVAL;
  return combo_nt3_launch(A, lda, Bimg, K, nullptr, mask, C, ldc, M, N, K, 0, g_products, 1, 0, 0, 0, nullptr, 0, stream);
}

extern "C" int combo_conv3x3_nhwc_x3_pre_f32(const float* X, long long ldx, const float* Wimg, const float* bias, float* Y,
                                             long long ldy, int B, int H, int W, int Cin, int Cout, int relu,
                                             combo_stream_t stream) {
  const long long M = (long long)B * H * W;
  if (!X || !Wimg || !Y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % kBK != 0 || ldx % 4 != 0 ||
      ((uintptr_t)X & 15) || ((uintptr_t)Wimg & 15) || M > 0x7fffffffLL / 4)
    return COMBO_EINVAL;
  const combo_nt3_conv cg{H, W, Cin, 0, 1, 1};
  return combo_nt3_launch(X, ldx, Wimg, 9LL * Cin, bias, nullptr, Y, ldy, M, Cout, 9 * Cin, relu, g_products, 1, 0, 0, 0, &cg, 0, stream);
}
