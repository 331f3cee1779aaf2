// Weight-streaming forward GEMM for a handful of rows: Y[M,N] = X[M,K] . W[N,K]^T (+ bias) (+ ReLU), M <= 64, exact fp32.
// The case is audio_mlp (reference: models/modeling/misc/audio_transformation.py:5-14: 128 -> 4096 -> 4096 -> 256 on the
// BT fused audio tokens): 73 MB of fp32 weights are read once per step for 40 rows - a GEMV regime.  A tiled GEMM leaves the
// chip idle (gemm_f32's 64 x 64 tiles: 64 workgroups, each grinding 131 K MFMA cycles over one weight panel); here the
// WEIGHT is the streamed operand and every CU takes an equal share of it:
//   * v_mfma_f32_16x16x4_f32 with A = 16 weight rows (output features), B = X^T (16 rows of X per tile, 1..4 tiles): a lane
//     reads 16 bytes (4 consecutive k) of its weight row per load - 1 KiB per wave instruction, straight to registers, 8
//     loads in flight per lane, no LDS round trip (GEMV rule: the streamed operand is used once per workgroup);
//   * MFMA step t takes element t of every lane's float4, i.e. k-slot kq <-> k = 4 kq + t of the 16-k group - X is loaded
//     with the same enumeration (X is 640 KB and lives in L2 / L1);
//   * balance: 12 MFMAs (3 row tiles x 4 steps) x 32 cycles per KiB of weights per wave = 10.7 B/clk/CU = 6.5 TB/s on 256 CUs,
//     i.e. the matrix pipe and HBM saturate together;
//   * grid = (N / 16 feature tiles) x (K splits): the 4 waves of a workgroup take a quarter of its K range each and are
//     summed through LDS; splits > 1 (the 4096 -> 256 layer: only 16 feature tiles) write partials that a small epilogue
//     kernel finishes (bias, ReLU) in a fixed order - deterministic, no atomics.
#include "combo_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MT, int NW>  // row tiles of 16: M <= 16 * MT; NW waves per workgroup share its K range (16: four per SIMD hide the
                           // HBM latency of the weight loads by themselves; 4 for short K)
__global__ void __launch_bounds__(NW * 64)
gemm_smallm_kernel(const float* __restrict__ X, long long ldx, const float* __restrict__ W, long long ldw,
                   const float* __restrict__ bias, float* __restrict__ Y, long long ldy, float* __restrict__ partial, int M, int N,
                   int K, int kchunk, int relu) {
  __shared__ f32x4 red[NW][MT][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * 16, split = blockIdx.y;
  const int kb = split * kchunk + wave * (kchunk / NW), ke = kb + kchunk / NW;  // this wave's k range (a multiple of 16 long)
  const float* wrow = W + (long long)min(n0 + i, N - 1) * ldw + 4 * kq;
  const float* xrow[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) xrow[t] = X + (long long)min(16 * t + i, M - 1) * ldx + 4 * kq;
  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int U = 8;  // 16-k groups per batch of weight loads
  for (int k0 = kb; k0 < ke; k0 += 16 * U) {
    f32x4 wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(k0 + 16 * u, ke - 16);  // (the last batch of a short range re-reads its final group; masked below)
      wv[u] = *reinterpret_cast<const f32x4*>(wrow + k);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k0 + 16 * u >= ke) break;
      f32x4 xv[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) xv[t] = *reinterpret_cast<const f32x4*>(xrow[t] + k0 + 16 * u);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][e], xv[t][e], acc[t], 0, 0, 0);
    }
  }
  // D[n = 4 kq + r][m = i]: sum the 4 waves through LDS, then lane (m, kq) owns 4 consecutive features of row m
#pragma unroll
  for (int t = 0; t < MT; ++t) red[wave][t][lane] = acc[t];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      f32x4 s = red[0][t][lane];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const f32x4 o = red[w][t][lane];
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
      }
      const int m = 16 * t + i, n = n0 + 4 * kq;
      if (m >= M) continue;
      float v[4] = {s.x, s.y, s.z, s.w};
      if (partial) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N) partial[((long long)split * M + m) * N + n + r] = v[r];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= N) continue;
          float y = v[r] + (bias ? bias[n + r] : 0.f);
          if (relu) y = fmaxf(y, 0.f);
          Y[(long long)m * ldy + n + r] = y;
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256)
smallm_finish_kernel(const float* __restrict__ partial, int splits, const float* __restrict__ bias, float* __restrict__ Y,
                     long long ldy, int M, int N, int relu) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= M * N) return;
  const int m = t / N, n = t - m * N;
  float s = bias ? bias[n] : 0.f;
  for (int z = 0; z < splits; ++z) s += partial[((long long)z * M + m) * N + n];
  Y[(long long)m * ldy + n] = relu ? fmaxf(s, 0.f) : s;
}

}  // namespace

extern "C" int combo_gemm_smallm_splits(int M, int N, int K) {
  // enough workgroups to give every CU a share of the weight stream; every wave needs a k range that is a multiple of 16
  (void)M;
  int n_cu = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
  const int n_tiles = (N + 15) / 16;
  int splits = 1;
  while (n_tiles * splits < n_cu && K % (splits * 2 * 64) == 0 && K / (splits * 2) >= 64) splits *= 2;
  return splits;
}

extern "C" int combo_gemm_smallm_f32(const float* X, long long ldx, const float* W, long long ldw, const float* bias, float* Y,
                                     long long ldy, float* partial_ws, int splits, int M, int N, int K, int relu,
                                     combo_stream_t stream) {
  if (!X || !W || !Y || M <= 0 || M > 64 || N <= 0 || K <= 0 || splits <= 0 || K % (64 * splits) != 0 || ldx % 4 != 0 || ldw % 4 != 0 ||
      (((uintptr_t)X | (uintptr_t)W) & 15) || (splits > 1 && !partial_ws))
    return COMBO_EINVAL;
  const dim3 grid((N + 15) / 16, splits);
  const int kchunk = K / splits;
  float* part = splits > 1 ? partial_ws : nullptr;
  const int mt = (M + 15) / 16;
  const bool wide = kchunk % 256 == 0;  // 16 waves per workgroup when every wave still gets whole 16-k groups
#define COMBO_SMALLM(MT)                                                                                                          \
  do {                                                                                                                            \
    if (wide)                                                                                                                     \
      hipLaunchKernelGGL((gemm_smallm_kernel<MT, 16>), grid, dim3(1024), 0, (hipStream_t)stream, X, ldx, W, ldw, bias, Y, ldy, part, \
                         M, N, K, kchunk, relu);                                                                                  \
    else                                                                                                                          \
      hipLaunchKernelGGL((gemm_smallm_kernel<MT, 4>), grid, dim3(256), 0, (hipStream_t)stream, X, ldx, W, ldw, bias, Y, ldy, part,   \
                         M, N, K, kchunk, relu);                                                                                  \
  } while (0)
  if (mt == 1) COMBO_SMALLM(1);
  else if (mt == 2) COMBO_SMALLM(2);
  else if (mt == 3) COMBO_SMALLM(3);
  else COMBO_SMALLM(4);
#undef COMBO_SMALLM
  if (splits > 1)
    hipLaunchKernelGGL(smallm_finish_kernel, dim3((M * N + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, splits, bias, Y, ldy, M, N, relu);
  return (int)hipGetLastError();
}


// ------------------------------------------------------------------------------------------------------------------
// C[M,N] = A[M,K] . W[K,N] for a tiny reduction length K <= 16 (the input gradient of class_embed: dY [BT*Q, K+1 classes] .
// W [K+1, 256], transformer_decoder.py:495): an outer-product sum per output, exact fp32 FMAs.  The BLAS library ran this
// as a 32x16x32 tile GEMM at 65 us per call (10 calls per step); a thread here computes 4 consecutive n of one row.
namespace {
__global__ void __launch_bounds__(256)
gemm_smallk_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ W, long long ldw, float* __restrict__ C,
                   long long ldc, long long M, int N4, int K) {
  const long long t = blockIdx.x * 256LL + threadIdx.x;
  if (t >= M * N4) return;
  const long long m = t / N4;
  const int n4 = (int)(t - m * N4);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < K; ++k) {
    const float a = A[m * lda + k];
    const float4 w = *reinterpret_cast<const float4*>(W + (long long)k * ldw + 4 * n4);
    acc.x = fmaf(a, w.x, acc.x); acc.y = fmaf(a, w.y, acc.y); acc.z = fmaf(a, w.z, acc.z); acc.w = fmaf(a, w.w, acc.w);
  }
  *reinterpret_cast<float4*>(C + m * ldc + 4 * n4) = acc;
}
}  // namespace

extern "C" int combo_gemm_smallk_f32(const float* A, long long lda, const float* W, long long ldw, float* C, long long ldc,
                                     long long M, int N, int K, combo_stream_t stream) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || K > 16 || N % 4 != 0 || ldw % 4 != 0 || ldc % 4 != 0 ||
      (((uintptr_t)W | (uintptr_t)C) & 15))
    return COMBO_EINVAL;
  const long long n = M * (N / 4);
  hipLaunchKernelGGL(gemm_smallk_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A, lda, W, ldw, C, ldc,
                     M, N / 4, K);
  return (int)hipGetLastError();
}


// dW[N,K] = dY[M,N]^T . X[M,K] (+ db[N] = sum_m dY) for a tiny output-row count N <= 16 (class_embed's weight gradient: N =
// classes + 1, M = BT*Q tokens, K = 256): one wave per token slice accumulates N x 4 outputs per lane in exact fp32 FMAs and
// writes a split-K partial; combo_splitk_reduce_f32 sums the slices.  (BLAS: a 32x16x32 tile GEMM, 65 us per call.)
namespace {
__global__ void __launch_bounds__(64)
gemm_tn_smalln_kernel(const float* __restrict__ dY, long long ldy, const float* __restrict__ X, long long ldx, long long M, int N,
                      int K4, int rows_per_slice, float* __restrict__ partials, float* __restrict__ db_partials) {
  const int slice = blockIdx.y, t = blockIdx.x * 64 + threadIdx.x;
  const long long m0 = (long long)slice * rows_per_slice, m1 = m0 + rows_per_slice < M ? m0 + rows_per_slice : M;
  const bool live = t < K4;
  float4 acc[16];
  float accb[16];
#pragma unroll
  for (int n = 0; n < 16; ++n) { acc[n] = make_float4(0.f, 0.f, 0.f, 0.f); accb[n] = 0.f; }
  for (long long mb = m0; mb < m1; mb += 8) {  // 8 rows per trip: their loads are in flight together (the loop is latency-bound)
    float4 x[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
      x[r] = (live && mb + r < m1) ? *reinterpret_cast<const float4*>(X + (mb + r) * ldx + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const long long m = mb + r < m1 ? mb + r : m1 - 1;
      const float keep = mb + r < m1 ? 1.f : 0.f;
#pragma unroll
      for (int n = 0; n < 16; ++n)
        if (n < N) {
          const float d = dY[m * ldy + n] * keep;  // wave-uniform address
          acc[n].x = fmaf(d, x[r].x, acc[n].x); acc[n].y = fmaf(d, x[r].y, acc[n].y);
          acc[n].z = fmaf(d, x[r].z, acc[n].z); acc[n].w = fmaf(d, x[r].w, acc[n].w);
          accb[n] += d;
        }
    }
  }
  if (live) {
#pragma unroll
    for (int n = 0; n < 16; ++n)
      if (n < N) *reinterpret_cast<float4*>(partials + ((long long)slice * N + n) * (4LL * K4) + 4 * t) = acc[n];
  }
  if (db_partials && t == 0) {
#pragma unroll
    for (int n = 0; n < 16; ++n)
      if (n < N) db_partials[(long long)slice * N + n] = accb[n];
  }
}
}  // namespace

extern "C" int combo_gemm_tn_smalln_slices(long long M) { return (int)((M + 31) / 32); }

extern "C" int combo_gemm_tn_smalln_f32(const float* dY, long long ldy, const float* X, long long ldx, long long M, int N, int K,
                                        float* partials, float* db_partials, combo_stream_t stream) {
  if (!dY || !X || !partials || M <= 0 || N <= 0 || N > 16 || K <= 0 || K % 4 != 0 || ldx % 4 != 0 ||
      (((uintptr_t)X | (uintptr_t)partials) & 15))
    return COMBO_EINVAL;
  const int slices = combo_gemm_tn_smalln_slices(M);
  hipLaunchKernelGGL(gemm_tn_smalln_kernel, dim3((unsigned)((K / 4 + 63) / 64), (unsigned)slices), dim3(64), 0, (hipStream_t)stream,
                     dY, ldy, X, ldx, M, N, K / 4, 32, partials, db_partials);
  return (int)hipGetLastError();
}
