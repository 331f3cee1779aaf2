// fp32-accurate dense layers on the bf16 matrix cores ("bf16x3"): Y[M,N] = X[M,K] . W[N,K]^T (+ bias, + ReLU).
//
// gfx950 has no TF32/xf32 path and its fp32-input MFMA runs at 1/16 of the bf16 rate (157 vs 2500 TFLOP/s).  The
// head's linears (K = 256..4096, M = BT*S = 41 160 tokens) are what the training step spends most of its time in
// when run through fp32 library GEMMs (~45 TFLOP/s measured).  Here every fp32 operand is split on the fly into
// hi = bf16(x) and lo = bf16(x - hi) and the product is accumulated in fp32 as
//        X.W^T  ~=  Xhi.Whi^T + Xhi.Wlo^T + Xlo.Whi^T           (the dropped Xlo.Wlo term is O(2^-16) relative)
// i.e. 3 bf16 MFMAs per fp32 MFMA-equivalent: 16/3 = 5.3x the fp32 matrix rate with ~1e-5 relative error (the
// parity tests hold the same tolerances as with fp32 library GEMMs).  Activations and weights stay fp32 in HBM; the
// split happens in registers between the global load and the LDS store, so no cast kernels and no extra traffic.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 2x2 MFMA 32x32x16
// tiles), BK = 32, operands staged as bf16 hi/lo planes in LDS (row stride 40 halfs = 80 B: conflict-free
// ds_read_b128 fragments), next tile's global loads are issued before the MFMAs of the current one.
// Operands may be "k-contiguous" ([rows][K], the forward and dX = dY.W with a transposed weight copy) or
// "row-contiguous" ([K][rows], used for dW = dY^T.X where the reduction runs over the token axis); the row-
// contiguous path transposes while storing to LDS.  Split-K over the reduction axis (grid.z) writes partial tiles
// that the caller sums (used by dW, whose output is tiny and whose reduction is 41 160 long).
#include "combo_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 128, BN = 128, BK = 32, LDK = 40;  // LDK: LDS row stride in halfs
constexpr int THREADS = 256;

// hi = x truncated to bf16 (so r = x - hi is exact in fp32 and |r| < 2^-7 |x|), lo = RNE_bf16(r): |x - hi - lo| <= 2^-16 |x|,
// unbiased.  3 VALU ops per element: v_and, v_sub, and half each of v_perm_b32 / v_cvt_pk_bf16_f32.
__device__ __forceinline__ unsigned pack_hi(float a, float b) {  // {bf16_trunc(a), bf16_trunc(b)} -> a in the low half
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float trunc_hi(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_rne(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned short f2bf_trunc(float x) { return (unsigned short)(__float_as_uint(x) >> 16); }
__device__ __forceinline__ unsigned short f2bf_rne(float x) { return (unsigned short)(pack_rne(x, 0.f) & 0xffffu); }

__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  hi = make_uint2(pack_hi(v.x, v.y), pack_hi(v.z, v.w));
  lo = make_uint2(pack_rne(v.x - trunc_hi(v.x), v.y - trunc_hi(v.y)), pack_rne(v.z - trunc_hi(v.z), v.w - trunc_hi(v.w)));
}

// One operand tile: `ROWC` = false: memory is [rows][K] (k contiguous), true: memory is [K][rows] (row contiguous).
// Global -> registers (4 x float4 per thread), then registers -> LDS hi/lo planes laid out [row][k].
template <bool ROWC>
struct TileLoader {
  float4 r[4];
  int rc_local;  // ROWC: first of this thread's 4 rows inside the tile (after clamping at the matrix edge)
  // Unconditional loads from clamped addresses (a branch per load would serialise them behind vmcnt(0) waits);
  // out-of-range ROWS only feed accumulators that are never stored, out-of-range K is zeroed with a select.
  __device__ __forceinline__ void load(const float* __restrict__ base, long long ld, int row0, int nrows, int k0, int kend,
                                       int tid) {
    if (!ROWC) {
      const int kc = (tid & 7) * 4;
      const bool kok = k0 + kc < kend;
      const int kcl = kok ? k0 + kc : kend - 4;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = min(row0 + (tid >> 3) + 32 * p, nrows - 1);
        const float4 v = *reinterpret_cast<const float4*>(base + (long long)row * ld + kcl);
        r[p] = kok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      const int rc = min(row0 + (tid & 31) * 4, nrows - 4);
      rc_local = rc - row0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int k = k0 + (tid >> 5) + 8 * p;
        const bool kok = k < kend;
        const float4 v = *reinterpret_cast<const float4*>(base + (long long)(kok ? k : kend - 1) * ld + rc);
        r[p] = kok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  __device__ __forceinline__ void store(unsigned short* __restrict__ shi, unsigned short* __restrict__ slo, int tid) {
    if (!ROWC) {
      const int kc = (tid & 7) * 4;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = (tid >> 3) + 32 * p;
        uint2 hi, lo;
        split4(r[p], hi, lo);
        *reinterpret_cast<uint2*>(shi + row * LDK + kc) = hi;
        *reinterpret_cast<uint2*>(slo + row * LDK + kc) = lo;
      }
    } else {
      const int rc = rc_local;  // may be negative only when the whole tile is out of range (never stored)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int k = (tid >> 5) + 8 * p;
        const float v[4] = {r[p].x, r[p].y, r[p].z, r[p].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          shi[(rc + j) * LDK + k] = f2bf_trunc(v[j]);
          slo[(rc + j) * LDK + k] = f2bf_rne(v[j] - trunc_hi(v[j]));
        }
      }
    }
  }
};

template <bool A_ROWC, bool B_ROWC>
__global__ void __launch_bounds__(THREADS)
gemm_x3_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ B, long long ldb,
               const float* __restrict__ bias, float* __restrict__ C, long long ldc, int M, int N, int K, int relu,
               int ksplit_len, long long split_stride) {
  __shared__ __attribute__((aligned(16))) unsigned short sAh[BM * LDK], sAl[BM * LDK], sBh[BN * LDK], sBl[BN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * ksplit_len, kend = min(K, kbeg + ksplit_len);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  TileLoader<A_ROWC> la;
  TileLoader<B_ROWC> lb;
  la.load(A, lda, m0, M, kbeg, kend, tid);
  lb.load(B, ldb, n0, N, kbeg, kend, tid);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();  // previous tile's fragments have been read
    la.store(sAh, sAl, tid);
    lb.store(sBh, sBl, tid);
    __syncthreads();
    if (k0 + BK < kend) {  // prefetch the next tile while the matrix cores work
      la.load(A, lda, m0, M, k0 + BK, kend, tid);
      lb.load(B, ldb, n0, N, k0 + BK, kend, tid);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ko = kk * 16 + (lane >> 5) * 8;
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + (lane & 31);
        ah[i] = *reinterpret_cast<const bf16x8*>(sAh + row * LDK + ko);
        al[i] = *reinterpret_cast<const bf16x8*>(sAl + row * LDK + ko);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wn * 64 + j * 32 + (lane & 31);
        bh[j] = *reinterpret_cast<const bf16x8*>(sBh + row * LDK + ko);
        bl[j] = *reinterpret_cast<const bf16x8*>(sBl + row * LDK + ko);
      }
      // small terms first, then the leading one; the 4 accumulators are independent between dependent MFMAs
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
  }
  // epilogue: D[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31]
  float* Cz = C + (long long)blockIdx.z * split_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < M) {
          float v = acc[i][j][e] + bv;
          if (relu) v = fmaxf(v, 0.f);
          Cz[(long long)row * ldc + col] = v;
        }
      }
    }
}

template <bool AR, bool BR>
int launch(const float* A, long long lda, const float* B, long long ldb, const float* bias, float* C, long long ldc, int M,
           int N, int K, int relu, int splits, long long split_stride, hipStream_t st) {
  int ks = (K + splits - 1) / splits;
  ks = (ks + BK - 1) / BK * BK;
  const int nz = (K + ks - 1) / ks;
  hipLaunchKernelGGL((gemm_x3_kernel<AR, BR>), dim3((N + BN - 1) / BN, (M + BM - 1) / BM, nz), dim3(THREADS), 0, st, A, lda,
                     B, ldb, bias, C, ldc, M, N, K, relu, ks, split_stride);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

// how many split-K partial tiles combo_gemm_x3_f32 will write for the given arguments (caller sizes C accordingly)
int combo_gemm_x3_splits(int K, int requested) {
  if (requested < 1) requested = 1;
  int ks = (K + requested - 1) / requested;
  ks = (ks + BK - 1) / BK * BK;
  return (K + ks - 1) / ks;
}

int combo_gemm_x3_f32(const float* A, long long lda, int a_rowc, const float* B, long long ldb, int b_rowc,
                      const float* bias, float* C, long long ldc, int M, int N, int K, int relu, int splits,
                      long long split_stride, combo_stream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return COMBO_EINVAL;
  // float4 loads: leading dimensions and K (k-contiguous) / M,N (row-contiguous) extents must be multiples of 4
  if ((lda & 3) || (ldb & 3) || ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15)) return COMBO_EINVAL;
  if ((!a_rowc && (K & 3)) || (!b_rowc && (K & 3)) || (a_rowc && (M & 3)) || (b_rowc && (N & 3))) return COMBO_EINVAL;
  if (splits > 1 && (bias || relu)) return COMBO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (!a_rowc && !b_rowc) return launch<false, false>(A, lda, B, ldb, bias, C, ldc, M, N, K, relu, splits, split_stride, st);
  if (!a_rowc && b_rowc) return launch<false, true>(A, lda, B, ldb, bias, C, ldc, M, N, K, relu, splits, split_stride, st);
  if (a_rowc && !b_rowc) return launch<true, false>(A, lda, B, ldb, bias, C, ldc, M, N, K, relu, splits, split_stride, st);
  return launch<true, true>(A, lda, B, ldb, bias, C, ldc, M, N, K, relu, splits, split_stride, st);
}

}  // extern "C"
