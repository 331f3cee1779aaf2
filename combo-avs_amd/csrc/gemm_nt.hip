// Forward / input-gradient GEMM of the head's dense layers with fp32 accuracy on the bf16 matrix cores:
//   C[M,N] = A[M,K] . B[N,K]^T (+ bias[N]) (+ ReLU)          A = tokens x features, B = nn.Linear weight
// (dX = dY . W is the same kernel with B = W^T, which the caller materialises once per step: it is a weight.)
//
// Both operands are K-contiguous, so an MFMA fragment (8 consecutive k of one row per lane) is a contiguous 32-byte read.
// Structure (the LDS-DMA ring of gemm_tn.hip v2, measured there at 2x hipBLASLt's 3xbf16 path):
//   * workgroup tile 256 (tokens) x 128 (n), 4 waves = 2 x 2, wave tile 128 x 64 = 4 x 2 MFMA 32x32 tiles (128 acc regs)
//   * per stage BK = 16 k: A 256 rows x 64 B + B 128 rows x 64 B = 24 KiB, streamed with global_load_lds_dwordx4 into a
//     3-deep ring (two workgroups per CU), one raw s_barrier per stage, counted s_waitcnt vmcnt, LDS reads as inline asm
//     (hipcc drains the DMA queue in front of any ds_read it can see)
//   * a 64-B row is four 16-B chunks; lane (row m, k-half g) reads chunks 2g, 2g+1.  Rows are 64 B apart, so the 16 rows
//     of a ds_read_b128 lane group would fall on 4 bank quads (4-way conflicts); the DMA therefore stores logical chunk c
//     of row r at physical chunk c ^ ((r >> 2) & 3) - a swizzle on the SOURCE address, the LDS image stays lane-linear -
//     which makes every lane group hit 16 distinct bank quads
//   * operands are split on the fly into bf16 hi/lo (x.w ~ hi.hi + hi.lo + lo.hi, error ~2^-16 relative)
//
// CONV = true turns the same kernel into the implicit-GEMM 3x3 / stride 1 / pad 1 convolution of the pixel decoder's FPN
// output layer (pixel_decoder/msdeformattn.py:281-286 in the reference; MIOpen's fp32 kernels were the largest library
// item of the head): A = the NHWC map as [B*H*W tokens, Cin], K = 9*Cin ordered (tap, cin), B = the weight as
// [Cout, 3, 3, Cin].  A BK = 16 stage lies inside one tap, so the only change is the DMA source row of the A pieces:
// token + dy*W + dx, or a row of zeros when the tap falls outside the map (pad).  dX is the same kernel on dY with the
// taps flipped and the weight transposed ([Cin, 3', 3', Cout]); the caller prepares both weight images once per step.
#include <cstdlib>

#include "combo_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float f4v __attribute__((ext_vector_type(4)));

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
__device__ __forceinline__ Frag make_frag(const f4v a, const f4v b) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned h[4], l[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    h[t] = pack_hi(v[2 * t], v[2 * t + 1]);
    l[t] = pack_rne(v[2 * t] - trunc_hi(v[2 * t]), v[2 * t + 1] - trunc_hi(v[2 * t + 1]));
  }
  Frag f;
  f.hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
  f.lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
  return f;
}
__device__ __forceinline__ f4v lds_read128(unsigned addr) {
  f4v r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void glds16(const float* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

constexpr int kBM = 256, kBN = 128, kBK = 16, kStages = 3;
constexpr int kABytes = kBM * kBK * 4, kBBytes = kBN * kBK * 4, kStage = kABytes + kBBytes;  // 16 + 8 = 24 KiB
constexpr int kPieces = kStage / 1024, kPPW = kPieces / 4;                                    // 24 pieces, 6 per wave

// DBG (ablation builds, env COMBO_NT_DBG, tools/bench_nt.py): 1 no epilogue stores, 2 no MFMA, 4 no bf16 split,
// 8 no LDS reads, 16 no DMA.  The product launches DBG = 0.
__device__ __attribute__((aligned(64))) float g_zero_row[16];  // zero-initialised: the source of padded taps

struct ConvGeom {
  int H, W, Cin;
};

template <int DBG, bool CONV>
__global__ void __launch_bounds__(256, 2)
gemm_nt_glds_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ B, long long ldb,
                    const float* __restrict__ bias, float* __restrict__ C, long long ldc, int M, int N, int K, int relu,
                    int remap, ConvGeom cg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // consecutive LOGICAL workgroups walk the n tiles of one token tile.  Hardware deals blockIdx round-robin over the 8
  // XCDs (private L2s), so the logical index gives every XCD a contiguous range: the n tiles that share the A rows of a
  // token tile run on ONE XCD at the same time and the rows come out of HBM once, not once per n tile.
  const int n_tiles = (N + kBN - 1) / kBN;
  const int logical = remap ? xcd_contiguous(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int m_blk = (logical / n_tiles) * kBM, n_blk = (logical % n_tiles) * kBN;
  const int nst = K / kBK;

  // DMA: piece q (1 KiB = 16 rows x 64 B) of a stage; lane -> (row, physical chunk); source = logical chunk (swizzle)
  const int p_row = lane >> 2, p_chunk = lane & 3;
  // CONV: this lane's four A rows (pieces wave, wave+4, wave+8, wave+12) -> 9-bit masks of the taps that stay inside the map
  unsigned tap_ok[4] = {0u, 0u, 0u, 0u};
  if (CONV) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = min(m_blk + (wave + 4 * u) * 16 + p_row, M - 1);
      const int x = t % cg.W, y = (t / cg.W) % cg.H;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        if (yy >= 0 && yy < cg.H && xx >= 0 && xx < cg.W) tap_ok[u] |= 1u << tap;
      }
    }
  }
  auto issue = [&](int s) {
    if (DBG & 16) return;
    char* st = smem + (s % kStages) * kStage;
    const int k0 = s * kBK;
    int tap = 0, cin0 = k0, shift = 0;  // CONV: the stage's tap (wave-uniform), first input channel and token shift
    if (CONV) {
      tap = k0 / cg.Cin;
      cin0 = k0 - tap * cg.Cin;
      shift = (tap / 3 - 1) * cg.W + (tap % 3 - 1);
    }
#pragma unroll
    for (int u = 0; u < kPPW; ++u) {
      const int q = wave + 4 * u;  // wave-uniform
      if (q < kABytes / 1024) {
        const int r = q * 16 + p_row;
        const int c = p_chunk ^ ((r >> 2) & 3);
        if (CONV) {
          const float* src = ((tap_ok[u & 3] >> tap) & 1u)
                                 ? A + (long long)(min(m_blk + r, M - 1) + shift) * lda + cin0 + c * 4
                                 : g_zero_row + c * 4;
          glds16(src, st + q * 1024);
        } else {
          glds16(A + (long long)min(m_blk + r, M - 1) * lda + k0 + c * 4, st + q * 1024);
        }
      } else {
        const int qb = q - kABytes / 1024;
        const int r = qb * 16 + p_row;
        const int c = p_chunk ^ ((r >> 2) & 3);
        glds16(B + (long long)min(n_blk + r, N - 1) * ldb + k0 + c * 4, st + kABytes + qb * 1024);
      }
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll
  for (int p = 0; p < kStages - 1; ++p)
    if (p < nst) issue(p);

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int m = lane & 31, g = lane >> 5;
  // byte offsets (within a stage) of this lane's two chunks for each of its A / B row tiles
  unsigned a_off[4][2], b_off[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 128 + i * 32 + m;
    const int s = (r >> 2) & 3;
    a_off[i][0] = (unsigned)(r * 64 + ((2 * g) ^ s) * 16);
    a_off[i][1] = (unsigned)(r * 64 + ((2 * g + 1) ^ s) * 16);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = wn * 64 + j * 32 + m;
    const int s = (r >> 2) & 3;
    b_off[j][0] = (unsigned)(kABytes + r * 64 + ((2 * g) ^ s) * 16);
    b_off[j][1] = (unsigned)(kABytes + r * 64 + ((2 * g + 1) ^ s) * 16);
  }

  for (int s = 0; s < nst; ++s) {
    if (DBG & 16) {}
    else if (nst - 1 - s >= 1) wait_vm<kPPW>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (s + kStages - 1 < nst) issue(s + kStages - 1);
    const unsigned so = lds0 + (unsigned)((s % kStages) * kStage);
    f4v ra[4][2], rb[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (DBG & 8) { ra[i][0] = ra[i][1] = (f4v){1.f + s, 2.f, 3.f, 4.f}; continue; }
      ra[i][0] = lds_read128(so + a_off[i][0]);
      ra[i][1] = lds_read128(so + a_off[i][1]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (DBG & 8) { rb[j][0] = rb[j][1] = (f4v){1.f, 2.f + s, 3.f, 4.f}; continue; }
      rb[j][0] = lds_read128(so + b_off[j][0]);
      rb[j][1] = lds_read128(so + b_off[j][1]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]),
                   "+v"(ra[3][0]), "+v"(ra[3][1]), "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1])
                 :
                 : "memory");
    Frag fa[4], fb[2];
    if (DBG & 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { fa[i].hi = __builtin_bit_cast(bf16x8, ra[i][0]); fa[i].lo = __builtin_bit_cast(bf16x8, ra[i][1]); }
#pragma unroll
      for (int j = 0; j < 2; ++j) { fb[j].hi = __builtin_bit_cast(bf16x8, rb[j][0]); fb[j].lo = __builtin_bit_cast(bf16x8, rb[j][1]); }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = make_frag(ra[i][0], ra[i][1]);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = make_frag(rb[j][0], rb[j][1]);
    }
    if (DBG & 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j][0] += __builtin_bit_cast(f4v, fa[i].lo).x + __builtin_bit_cast(f4v, fb[j].hi).y;
          acc[i][j][1] += __builtin_bit_cast(f4v, fa[i].hi).z + __builtin_bit_cast(f4v, fb[j].lo).w;
          acc[i][j][2] += __builtin_bit_cast(f4v, fa[i].lo).z + __builtin_bit_cast(f4v, fb[j].hi).w;
          acc[i][j][3] += __builtin_bit_cast(f4v, fa[i].hi).x + __builtin_bit_cast(f4v, fb[j].lo).y;
        }
      continue;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].lo, fb[j].hi, acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].hi, fb[j].lo, acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].hi, fb[j].hi, acc[i][j], 0, 0, 0);
  }

  // epilogue: D tile = 32 tokens x 32 n; lane holds n = lane & 31 and tokens (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n_blk + wn * 64 + j * 32 + m;
    if (n >= N) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m_blk + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
        if (row < M && (!(DBG & 1) || acc[i][j][e] == 1234.5f)) {
          float v = acc[i][j][e] + bv;
          if (relu) v = fmaxf(v, 0.f);
          C[(long long)row * ldc + n] = v;
        }
      }
  }
}

typedef void (*kern_t)(const float*, long long, const float*, long long, const float*, float*, long long, int, int, int, int,
                       int, ConvGeom);

}  // namespace

extern "C" int combo_gemm_nt_x3_f32(const float* A, long long lda, const float* B, long long ldb, const float* bias,
                                    float* C, long long ldc, int M, int N, int K, int relu, combo_stream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || K % kBK != 0 || lda % 4 != 0 || ldb % 4 != 0 ||
      ((uintptr_t)A & 15) || ((uintptr_t)B & 15))
    return COMBO_EINVAL;
  constexpr int lds = kStages * kStage;
  static const int dbg = [] { const char* e = getenv("COMBO_NT_DBG"); return e ? atoi(e) : 0; }();
  static const kern_t kern = dbg == 1 ? gemm_nt_glds_kernel<1, false> : dbg == 2 ? gemm_nt_glds_kernel<2, false>
                           : dbg == 3 ? gemm_nt_glds_kernel<3, false> : dbg == 4 ? gemm_nt_glds_kernel<4, false>
                           : dbg == 7 ? gemm_nt_glds_kernel<7, false> : dbg == 15 ? gemm_nt_glds_kernel<15, false>
                           : dbg == 31 ? gemm_nt_glds_kernel<31, false> : dbg == 30 ? gemm_nt_glds_kernel<30, false>
                           : dbg == 16 ? gemm_nt_glds_kernel<16, false> : gemm_nt_glds_kernel<0, false>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const long long tiles = (long long)((M + kBM - 1) / kBM) * ((N + kBN - 1) / kBN);
  if (tiles > 0x7fffffffLL) return COMBO_EINVAL;
  static const int remap = [] { const char* e = getenv("COMBO_GEMM_XCD"); return e ? atoi(e) : 1; }();
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, (hipStream_t)stream, A, lda, B, ldb, bias, C, ldc, M, N, K,
                     relu, remap, ConvGeom{1, 1, K});
  return (int)hipGetLastError();
}

extern "C" int combo_conv3x3_nhwc_x3_f32(const float* X, long long ldx, const float* Wm, const float* bias, float* Y,
                                         long long ldy, int B, int H, int W, int Cin, int Cout, int relu,
                                         combo_stream_t stream) {
  const long long M = (long long)B * H * W;
  if (!X || !Wm || !Y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % kBK != 0 || ldx % 4 != 0 ||
      ((uintptr_t)X & 15) || ((uintptr_t)Wm & 15) || M > 0x7fffffffLL / 4)
    return COMBO_EINVAL;
  constexpr int lds = kStages * kStage;
  static const kern_t kern = gemm_nt_glds_kernel<0, true>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const long long tiles = ((M + kBM - 1) / kBM) * ((Cout + kBN - 1) / kBN);
  if (tiles > 0x7fffffffLL) return COMBO_EINVAL;
  static const int remap = [] { const char* e = getenv("COMBO_GEMM_XCD"); return e ? atoi(e) : 1; }();
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, (hipStream_t)stream, X, ldx, Wm, (long long)9 * Cin, bias, Y,
                     ldy, (int)M, Cout, 9 * Cin, relu, remap, ConvGeom{H, W, Cin});
  return (int)hipGetLastError();
}
