// Weight-gradient GEMM of the head's dense layers: dW[N,K] = dY[M,N]^T . X[M,K], reduction over the M = BT*S tokens.
//
// hipBLASLt needs ~490 us for the encoder-FFN shapes (41160 x 1024 x 256: tiny output, 41 160-long reduction; every
// layout / split mode measured the same, tools/blas_test2.py), which made the dW GEMMs ~12 ms of the training step.
// Both operands are row-major with the REDUCTION index as the row, so an MFMA fragment (8 consecutive reduction
// elements per lane for a fixed output row/column) is a strided column walk: lanes of a wave read 32 consecutive
// floats of one row (a coalesced 128-B segment), 8 rows per fragment.  That access pattern needs no LDS staging at
// all: fragments are loaded straight from global/L2 into registers, split into bf16 hi/lo (fp32-accurate 3-product
// scheme of gemm_x3.hip) and fed to v_mfma_f32_32x32x16_bf16.  Split-K over the token axis fills the chip
// (16 output tiles x 64 splits for the FFN shape); the partial tiles are summed by the caller.
#include "combo_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ unsigned pack_hi(float a, float b) {
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float trunc_hi(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_rne(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

struct Frag {
  bf16x8 hi, lo;
};

__device__ __forceinline__ Frag make_frag(const float (&v)[8]) {
  Frag f;
  unsigned h[4], l[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    h[t] = pack_hi(v[2 * t], v[2 * t + 1]);
    l[t] = pack_rne(v[2 * t] - trunc_hi(v[2 * t]), v[2 * t + 1] - trunc_hi(v[2 * t + 1]));
  }
  f.hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
  f.lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
  return f;
}

// 8 rows (reduction index) x this lane's column, rows beyond `mend` read as zero
__device__ __forceinline__ void load8(const float* __restrict__ base, long long ld, int m, int mend, int mlast, float (&v)[8]) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int r = m + t;
    const float x = base[(long long)min(r, mlast) * ld];
    v[t] = r < mend ? x : 0.f;
  }
}

// grid: (K/128 tiles, N/128 tiles, splits); block: 256 threads = 4 waves (2 along n x 2 along k), 64x64 per wave
__global__ void __launch_bounds__(256)
gemm_tn_x3_kernel(const float* __restrict__ dY, long long ldy, const float* __restrict__ X, long long ldx,
                  float* __restrict__ out, float* __restrict__ db_part, int M, int N, int K, int mchunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = blockIdx.y * 128 + wn * 64, k0 = blockIdx.x * 128 + wk * 64;
  const int mbeg = blockIdx.z * mchunk, mend = min(M, mbeg + mchunk);
  const int col = lane & 31, kg = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // column pointers (clamped at the matrix edge: those accumulators are never stored)
  const float* pa[2] = {dY + min(n0 + col, N - 1), dY + min(n0 + 32 + col, N - 1)};
  const float* pb[2] = {X + min(k0 + col, K - 1), X + min(k0 + 32 + col, K - 1)};
  float va[2][8], vb[2][8];
  // bias gradient db[n] = sum_m dY[m,n] rides along: the waves of the first K tile already hold every dY element
  const bool do_db = db_part != nullptr && blockIdx.x == 0 && wk == 0;
  float colsum[2] = {0.f, 0.f};
  int m = mbeg + kg * 8;
#pragma unroll
  for (int i = 0; i < 2; ++i) { load8(pa[i], ldy, m, mend, M - 1, va[i]); load8(pb[i], ldx, m, mend, M - 1, vb[i]); }
  for (; m - kg * 8 < mend; m += 16) {
    Frag fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { fa[i] = make_frag(va[i]); fb[i] = make_frag(vb[i]); }
    if (do_db) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        colsum[i] += ((va[i][0] + va[i][1]) + (va[i][2] + va[i][3])) + ((va[i][4] + va[i][5]) + (va[i][6] + va[i][7]));
    }
    if (m - kg * 8 + 16 < mend) {  // next 16 reduction rows, in flight during the MFMAs
#pragma unroll
      for (int i = 0; i < 2; ++i) { load8(pa[i], ldy, m + 16, mend, M - 1, va[i]); load8(pb[i], ldx, m + 16, mend, M - 1, vb[i]); }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].lo, fb[j].hi, acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].hi, fb[j].lo, acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].hi, fb[j].hi, acc[i][j], 0, 0, 0);
  }
  if (do_db) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float t = colsum[i] + __shfl_xor(colsum[i], 32);
      const int nr = n0 + i * 32 + col;
      if (kg == 0 && nr < N) db_part[(long long)blockIdx.z * N + nr] = t;
    }
  }
  float* o = out + (long long)blockIdx.z * N * K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kc = k0 + j * 32 + col;
      if (kc >= K) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int nr = n0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
        if (nr < N) o[(long long)nr * K + kc] = acc[i][j][e];
      }
    }
}

}  // namespace

extern "C" {

int combo_gemm_tn_splits(int M, int N, int K) {
  const long long tiles = (long long)((N + 127) / 128) * ((K + 127) / 128);
  long long s = (2048 + tiles - 1) / tiles;           // ~8 workgroups per CU
  const long long maxs = (M + 255) / 256;             // at least 256 reduction rows per split
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  return (int)s;
}

int combo_gemm_tn_x3_f32(const float* dY, long long ldy, const float* X, long long ldx, float* out_partials,
                         float* db_partials, int M, int N, int K, int splits, combo_stream_t stream) {
  if (!dY || !X || !out_partials || M <= 0 || N <= 0 || K <= 0 || splits <= 0) return COMBO_EINVAL;
  int mchunk = (M + splits - 1) / splits;
  mchunk = (mchunk + 15) / 16 * 16;
  const int nz = (M + mchunk - 1) / mchunk;
  if (nz != splits) return COMBO_EINVAL;  // caller sizes `out_partials` with combo_gemm_tn_splits / this rounding
  hipLaunchKernelGGL(gemm_tn_x3_kernel, dim3((K + 127) / 128, (N + 127) / 128, nz), dim3(256), 0, (hipStream_t)stream, dY,
                     ldy, X, ldx, out_partials, db_partials, M, N, K, mchunk);
  return (int)hipGetLastError();
}

}  // extern "C"
