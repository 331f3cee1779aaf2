// gemm_nt v2: C[M,N] = A[M,K] . B[N,K]^T (+ bias) (+ ReLU), fp32 accuracy on the bf16 matrix cores (3-way split), and the
// implicit-GEMM 3x3 convolution built on it (CONV = true, see gemm_nt.hip for the addressing).
//
// Every change against gemm_nt.hip (v1) answers one line of v1's ablation on MI355X (tools/abl_nt.py, 41160 x 256 -> 1024:
// 141 us = 18 launch/loop + 23 LDS-DMA + 5 LDS reads + 5 split + 43 MFMA + 49 epilogue, i.e. NOTHING overlapped):
//   * persistent workgroups (grid = 2 per CU) walking the tile list: no workgroup launch / teardown between tiles and the
//     first stages of the NEXT tile are already in flight while the epilogue of the current one is stored (an optional
//     half-tile start stagger of the second workgroup per CU, COMBO_NT2_STAGGER=1, measured neutral to -3 us: off)
//   * the weight operand arrives PRE-SPLIT (combo_presplit_bf16x2_f32: per 8 k a 16-B bf16 `hi` group and a 16-B `lo`
//     group, the same 4 bytes per element): v1 re-split every weight row in every workgroup (161 x 2 times)
//   * waves 4 x 1 instead of 2 x 2: a wave owns 64 token rows x all 128 columns, so an A row is split once per workgroup
//     instead of twice: 48 conversion VALU ops per stage instead of 144 (24 MFMAs)
//   * LDS addresses: one base per operand chunk + immediate offsets (the swizzle only depends on the row's bits 2-3)
// Measured (tools/abl_nt.py, tools/clock_probe.py): bit-identical to v1, 5-8 % faster (41160x256->1024: 144 -> 132 us,
// 125440x2304->256: 576 -> 534 us = 831 TF/s of bf16 MFMA work).  The chip is POWER-bound here, not issue-bound: both
// kernels pull the 1400 W package limit with sclk falling to ~1.9 GHz (hipBLASLt's plain bf16 GEMM: 1374 TF/s at 8192^3,
// 1092 TF/s on the 125440x2304x256 shape, 479 TF/s at 41160x256->1024 where this kernel does 3x the MFMA work in 2.9x the
// time), which is why the phases of v1's ablation add up instead of overlapping: at the cap, time follows energy.
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
__device__ __forceinline__ void split8(const f4v a, const f4v b, bf16x8& hi, bf16x8& lo) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned h[4], l[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    h[t] = pack_hi(v[2 * t], v[2 * t + 1]);
    l[t] = pack_rne(v[2 * t] - trunc_hi(v[2 * t]), v[2 * t + 1] - trunc_hi(v[2 * t + 1]));
  }
  hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
  lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}
// LDS reads as inline asm: hipcc drains the LDS-DMA queue (s_waitcnt vmcnt(0)) in front of any ds_read it can see
template <int OFF>
__device__ __forceinline__ f4v lds_read128(unsigned addr) {
  f4v r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
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
constexpr int kPPW = kStage / 1024 / 4;                                                       // 6 DMA pieces per wave

__device__ __attribute__((aligned(64))) float g_zero_row2[16];  // zero-initialised: the source of padded taps

struct ConvGeom2 {
  int H, W, Cin;
};

template <bool CONV>
__global__ void __launch_bounds__(256, 2)
gemm_nt2_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ Bimg, long long ldb,
                const float* __restrict__ bias, float* __restrict__ C, long long ldc, int M, int N, int K, int relu,
                int stagger, ConvGeom2 cg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_tiles = (N + kBN - 1) / kBN;
  const int tiles = ((M + kBM - 1) / kBM) * n_tiles;
  const int G = gridDim.x;
  // logical workgroup index: every XCD owns a contiguous range, so the n tiles sharing the A rows of a token tile run on
  // one XCD (one L2) in the same round
  const int w = xcd_contiguous(blockIdx.x, G);
  const int nst = K / kBK;
  if (stagger > 0 && (int)blockIdx.x >= G / 2) {  // the second workgroup of each CU starts half a tile late
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(32);
  }

  // ---------------- issue cursor: (tile, stage) of the next stage to stream into the ring ----------------
  const int p_row = lane >> 2, p_chunk = lane & 3;
  int i_tile = w, i_s = 0, i_slot = 0, issued = 0;
  int i_m_blk = 0, i_n_blk = 0, i_tap = 0, i_cin0 = 0;
  unsigned tap_ok[4] = {0u, 0u, 0u, 0u};  // CONV: this lane's four A rows -> 9-bit masks of the taps inside the map
  auto open_tile = [&]() {
    i_m_blk = (i_tile / n_tiles) * kBM;
    i_n_blk = (i_tile % n_tiles) * kBN;
    i_s = 0; i_tap = 0; i_cin0 = 0;
    if (CONV) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = min(i_m_blk + (wave + 4 * u) * 16 + p_row, M - 1);
        const int x = t % cg.W, y = (t / cg.W) % cg.H;
        unsigned ok = 0u;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
          if (yy >= 0 && yy < cg.H && xx >= 0 && xx < cg.W) ok |= 1u << tap;
        }
        tap_ok[u] = ok;
      }
    }
  };
  auto issue_next = [&]() {
    if (i_tile >= tiles) return;
    char* st = smem + i_slot * kStage;
    const int k0 = i_s * kBK;
    const int shift = CONV ? (i_tap / 3 - 1) * cg.W + (i_tap % 3 - 1) : 0;
#pragma unroll
    for (int u = 0; u < kPPW; ++u) {
      const int q = wave + 4 * u;  // wave-uniform piece (1 KiB = 16 rows x 64 B); q < 16: A rows, else B rows
      if (u < 4) {
        const int r = q * 16 + p_row;
        const int c = p_chunk ^ ((r >> 2) & 3);  // swizzle on the SOURCE chunk, the LDS image stays lane-linear
        const float* src;
        if (CONV)
          src = ((tap_ok[u & 3] >> i_tap) & 1u) ? A + (long long)(min(i_m_blk + r, M - 1) + shift) * lda + i_cin0 + c * 4
                                                : g_zero_row2 + c * 4;
        else
          src = A + (long long)min(i_m_blk + r, M - 1) * lda + k0 + c * 4;
        glds16(src, st + q * 1024);
      } else {
        const int qb = q - 16;
        const int r = qb * 16 + p_row;
        const int c = p_chunk ^ ((r >> 2) & 3);
        glds16(Bimg + (long long)min(i_n_blk + r, N - 1) * ldb + k0 + c * 4, st + kABytes + qb * 1024);
      }
    }
    ++issued;
    i_slot = i_slot == kStages - 1 ? 0 : i_slot + 1;
    ++i_s;
    if (CONV) {
      i_cin0 += kBK;
      if (i_cin0 == cg.Cin) { i_cin0 = 0; ++i_tap; }
    }
    if (i_s == nst) {
      i_tile += G;
      if (i_tile < tiles) open_tile();
    }
  };
  if (i_tile < tiles) open_tile();
  issue_next();
  issue_next();

  // ---------------- LDS read addresses: lane (row m, k-half g) reads chunks 2g, 2g+1 of its rows ----------------
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int m = lane & 31, g = lane >> 5;
  const int sw = (m >> 2) & 3;  // rows wave*64 + i*32 + m and j*32 + m all share the swizzle of m
  const unsigned a_c0 = lds0 + (unsigned)((wave * 64 + m) * 64 + ((2 * g) ^ sw) * 16);
  const unsigned a_c1 = lds0 + (unsigned)((wave * 64 + m) * 64 + ((2 * g + 1) ^ sw) * 16);
  const unsigned b_c0 = lds0 + (unsigned)(kABytes + m * 64 + ((2 * g) ^ sw) * 16);
  const unsigned b_c1 = lds0 + (unsigned)(kABytes + m * 64 + ((2 * g + 1) ^ sw) * 16);

  int consumed = 0, c_slot = 0;
  for (int tile = w; tile < tiles; tile += G) {
    const int m_blk = (tile / n_tiles) * kBM, n_blk = (tile % n_tiles) * kBN;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    for (int s = 0; s < nst; ++s) {
      // the stage to consume has landed once every OLDER vector-memory operation of this wave is done.  Inside a tile
      // only ring loads are pending (counted wait: one younger stage may stay in flight); the first stage of a tile also
      // drains the epilogue stores of the previous tile (stores and loads share vmcnt and may retire out of order).
      if (s == 0 || issued - consumed == 1) wait_vm<0>();
      else wait_vm<kPPW>();
      __builtin_amdgcn_s_barrier();  // everybody's pieces landed; everybody finished reading the slot refilled below
      issue_next();
      const unsigned so = (unsigned)(c_slot * kStage);
      f4v ra[2][2];
      bf16x8 bh[4], bl[4];
      ra[0][0] = lds_read128<0>(a_c0 + so);
      ra[0][1] = lds_read128<0>(a_c1 + so);
      ra[1][0] = lds_read128<2048>(a_c0 + so);
      ra[1][1] = lds_read128<2048>(a_c1 + so);
      f4v rb0[4], rb1[4];
      rb0[0] = lds_read128<0>(b_c0 + so);    rb1[0] = lds_read128<0>(b_c1 + so);
      rb0[1] = lds_read128<2048>(b_c0 + so); rb1[1] = lds_read128<2048>(b_c1 + so);
      rb0[2] = lds_read128<4096>(b_c0 + so); rb1[2] = lds_read128<4096>(b_c1 + so);
      rb0[3] = lds_read128<6144>(b_c0 + so); rb1[3] = lds_read128<6144>(b_c1 + so);
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(rb0[0]), "+v"(rb1[0]),
                     "+v"(rb0[1]), "+v"(rb1[1]), "+v"(rb0[2]), "+v"(rb1[2]), "+v"(rb0[3]), "+v"(rb1[3])
                   :
                   : "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bh[j] = __builtin_bit_cast(bf16x8, rb0[j]);
        bl[j] = __builtin_bit_cast(bf16x8, rb1[j]);
      }
      bf16x8 ah[2], al[2];
      split8(ra[0][0], ra[0][1], ah[0], al[0]);
      split8(ra[1][0], ra[1][1], ah[1], al[1]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      ++consumed;
      c_slot = c_slot == kStages - 1 ? 0 : c_slot + 1;
    }

    // epilogue: D tile = 32 tokens x 32 n; lane holds n = lane & 31 and tokens (e&3) + 8*(e>>2) + 4*(lane>>5).  The
    // stores drain while the next tile's first stages (already in flight) land.
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n_blk + j * 32 + m;
      if (n >= N) continue;
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m_blk + wave * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
          if (row < M) {
            float v = acc[i][j][e] + bv;
            if (relu) v = fmaxf(v, 0.f);
            C[(long long)row * ldc + n] = v;
          }
        }
    }
  }
}

// Weight image for gemm_nt2: element (n, k) = src[n*ld_row + k*ld_col]; per 8 consecutive k a 16-B group of bf16 `hi`
// (truncated) followed by a 16-B group of bf16 `lo` = rne(x - hi): 4 bytes per element, row stride K floats.
__global__ void __launch_bounds__(256)
presplit_kernel(const float* __restrict__ src, long long ld_row, long long ld_col, int N, int K, uint4* __restrict__ img) {
  const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int kg = K >> 3;
  if (t >= (long long)N * kg) return;
  const int n = (int)(t / kg), g8 = (int)(t - (long long)n * kg);
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = src[(long long)n * ld_row + (long long)(g8 * 8 + i) * ld_col];
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = pack_hi(v[2 * i], v[2 * i + 1]);
    l[i] = pack_rne(v[2 * i] - trunc_hi(v[2 * i]), v[2 * i + 1] - trunc_hi(v[2 * i + 1]));
  }
  img[t * 2] = make_uint4(h[0], h[1], h[2], h[3]);
  img[t * 2 + 1] = make_uint4(l[0], l[1], l[2], l[3]);
}

template <bool CONV>
int launch_nt2(const float* A, long long lda, const float* Bimg, const float* bias, float* C, long long ldc, long long M, int N,
               int K, int relu, ConvGeom2 cg, combo_stream_t stream) {
  constexpr int lds = kStages * kStage;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt2_kernel<CONV>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const long long tiles = ((M + kBM - 1) / kBM) * ((N + kBN - 1) / kBN);
  if (tiles > 0x7fffffffLL) return COMBO_EINVAL;
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return cus > 0 ? cus : 256;
  }();
  static const int stagger_on = [] { const char* e = getenv("COMBO_NT2_STAGGER"); return e ? atoi(e) : 0; }();
  const int grid = (int)(tiles < 2LL * n_cu ? tiles : 2LL * n_cu);
  // half a tile of the main loop, in units of s_sleep(32) = 2048 cycles (a BK = 16 stage costs ~1500 cycles per wave)
  int stagger = 0;
  if (stagger_on && tiles > n_cu) {
    stagger = (K / kBK) * 750 / 2048;
    if (stagger < 1) stagger = 1;
    if (stagger > 8) stagger = 8;
  }
  hipLaunchKernelGGL(gemm_nt2_kernel<CONV>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, A, lda, Bimg,
                     (long long)K, bias, C, ldc, (int)M, N, K, relu, stagger, cg);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int combo_presplit_bf16x2_f32(const float* src, long long ld_row, long long ld_col, int N, int K, float* img,
                                         combo_stream_t stream) {
  if (!src || !img || N <= 0 || K <= 0 || K % 8 != 0 || ((uintptr_t)img & 15)) return COMBO_EINVAL;
  const long long threads = (long long)N * (K / 8);
  hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld_row,
                     ld_col, N, K, reinterpret_cast<uint4*>(img));
  return (int)hipGetLastError();
}

extern "C" int combo_gemm_nt_x3_pre_f32(const float* A, long long lda, const float* Bimg, const float* bias, float* C,
                                        long long ldc, int M, int N, int K, int relu, combo_stream_t stream) {
  if (!A || !Bimg || !C || M <= 0 || N <= 0 || K <= 0 || K % kBK != 0 || lda % 4 != 0 || ((uintptr_t)A & 15) ||
      ((uintptr_t)Bimg & 15))
    return COMBO_EINVAL;
  return launch_nt2<false>(A, lda, Bimg, bias, C, ldc, M, N, K, relu, ConvGeom2{1, 1, K}, stream);
}

extern "C" int combo_conv3x3_nhwc_x3_pre_f32(const float* X, long long ldx, const float* Wimg, const float* bias, float* Y,
                                             long long ldy, int B, int H, int W, int Cin, int Cout, int relu,
                                             combo_stream_t stream) {
  const long long M = (long long)B * H * W;
  if (!X || !Wimg || !Y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % kBK != 0 || ldx % 4 != 0 ||
      ((uintptr_t)X & 15) || ((uintptr_t)Wimg & 15) || M > 0x7fffffffLL / 4)
    return COMBO_EINVAL;
  return launch_nt2<true>(X, ldx, Wimg, bias, Y, ldy, M, Cout, 9 * Cin, relu, ConvGeom2{H, W, Cin}, stream);
}
