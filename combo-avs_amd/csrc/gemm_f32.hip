// gemm_nt_f32: C[M,N] = A[M,K] . B[N,K]^T (+ bias) (+ ReLU) in EXACT fp32 on the matrix cores
// (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, one rounding per product - bitwise an fmaf chain), and the implicit-GEMM
// 3x3 convolution on the same loop (CONV = true).
//
// Why this kernel exists (round 2): the 3-product bf16 split of gemm_nt2.hip carries ~2^-17 relative error per product
// against fp32's 2^-24.  Every FORWARD dense layer of the head ends, a few layers later, in `sigmoid(logit) < 0.5` (the
// attention masks of the decoder, transformer_decoder.py:502-507): a cell whose logit lies within the error band of 0
// flips, and the flip perturbs that query in all following layers - 0.07 / 0.42 / 0.66 % of the mask logits of prediction
// heads 7 / 8 / 9 ended beyond the north-star's 1e-3 tolerance.  With true-fp32 GEMMs the same head has no outlier
// (tests/test_head_gpu.py).  So the forward path computes in fp32; the gradient GEMMs (no threshold downstream, tolerance
// 2e-3) keep the 3-product split, which is ~1.9x faster.
//
// Roofline: MFMA-bound.  The f32 MFMA issues once per 64 cycles per SIMD = 64 flop/clk/SIMD = 157.3 TFLOP/s on the chip
// (the fp32 vector rate, 1/16 of bf16).  Per wave and BK = 16 stage a 2 x 4 tile set is 64 MFMAs = 4096 cycles against
// 12 ds_read_b128 and 6 LDS-DMA pieces: the operand traffic is a rounding error, so the design keeps gemm_nt2's proven
// streaming skeleton unchanged (persistent workgroups, XCD-contiguous tile order, LDS-DMA ring with the chunk swizzle on
// the SOURCE address, counted vmcnt across a raw s_barrier, buffer-store epilogue) and spends its thought on tile
// quantisation instead - at 64 cycles per MFMA an idle SIMD is the only way to lose time:
//   wide   256 x 128 (4 x 1 waves, 2 x 4 tiles, 2 workgroups/CU)  large M, N >= 128
//   mid    128 x 128 (2 x 2 waves, 2 x 2 tiles, 3 workgroups/CU)  M ~ 100 per batch entry (mask logits), mid-sized layers
//   skinny  64 x  64 (2 x 2 waves, 1 x 1 tiles, ring = a K = 256 panel) the decoder's 4000-token layers: 252 tiles
// chosen per call by the smallest "busiest CU" load (launch_f32 below).
//
// k order inside a stage: lane (row m, half g) holds k = 8g .. 8g+7 of its rows (two 16-byte chunks); MFMA step t pairs
// k = t (g = 0) with k = 8 + t (g = 1) for both operands - any pairing is a valid contraction order.
#include <cstdlib>
#include <type_traits>

#include "combo_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float f4v __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ f4v lds_read128(unsigned addr) {  // inline asm: hipcc drains the LDS-DMA queue before a visible ds_read
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

constexpr int kBK = 16;

__device__ __attribute__((aligned(64))) float g_zero_row_f32[16];  // zero-initialised: the source of padded conv taps

struct ConvGeomF {
  int H, W, Cin;
};

template <int WM_, int WN_, int TI_, int TJ_, int ST_, int WGS_>
struct F32Cfg {
  static constexpr int WM = WM_, WN = WN_, TI = TI_, TJ = TJ_, ST = ST_, WGS = WGS_;  // WGS: resident workgroups per CU
  static constexpr int BM = WM * TI * 32, BN = WN * TJ * 32;
  static constexpr int A_BYTES = BM * kBK * 4, B_BYTES = BN * kBK * 4, STAGE = A_BYTES + B_BYTES;
  static constexpr int A_PIECES = A_BYTES / 1024, PIECES = STAGE / 1024, PPW = PIECES / 4, APW = A_PIECES / 4;
  static constexpr int LDS = ST * STAGE;
  static_assert(WM * WN == 4 && PIECES % 4 == 0 && A_PIECES % 4 == 0, "4 waves share the DMA pieces evenly");
  static_assert((ST - 1) * PPW <= 63, "vmcnt is a 6-bit counter");
  static_assert(LDS * WGS <= 160 * 1024, "LDS budget of a CU");
};
typedef F32Cfg<4, 1, 2, 4, 3, 2> FWide;
typedef F32Cfg<2, 2, 2, 2, 3, 3> FMid;
typedef F32Cfg<2, 2, 1, 1, 16, 1> FSkinny;

template <int PPW, int ST>
__device__ __forceinline__ void wait_younger(int younger) {  // s_waitcnt vmcnt(younger * PPW): the immediate must be static
  if constexpr (ST <= 3) {
    if (younger == 0) wait_vm<0>();
    else wait_vm<PPW>();
    return;
  }
  switch (younger) {
    case 0: wait_vm<0>(); break;
    case 1: wait_vm<PPW>(); break;
    case 2: wait_vm<2 * PPW>(); break;
    case 3: wait_vm<3 * PPW>(); break;
    case 4: wait_vm<4 * PPW>(); break;
    case 5: wait_vm<5 * PPW>(); break;
    case 6: wait_vm<6 * PPW>(); break;
    case 7: wait_vm<7 * PPW>(); break;
    case 8: wait_vm<(8 * PPW) & 63>(); break;
    case 9: wait_vm<(9 * PPW) & 63>(); break;
    case 10: wait_vm<(10 * PPW) & 63>(); break;
    case 11: wait_vm<(11 * PPW) & 63>(); break;
    case 12: wait_vm<(12 * PPW) & 63>(); break;
    case 13: wait_vm<(13 * PPW) & 63>(); break;
    case 14: wait_vm<(14 * PPW) & 63>(); break;
    default: wait_vm<(15 * PPW) & 63>(); break;
  }
}

struct F32Args {
  const float* A; long long lda;
  const float* B; long long ldb;
  const float* bias;      // [N] or nullptr
  float* C; long long ldc;
  int M, N, K, relu, c_bytes, batch;
  long long sA, sB, sC;   // batch strides (elements)
  ConvGeomF cg;
  unsigned long long* ts; // device-side timing slot (combo_common.h) or nullptr
};

template <bool CONV, typename Cfg>
__global__ void __launch_bounds__(256, Cfg::WGS)
gemm_nt_f32_kernel(const F32Args p) {
  constexpr int BM = Cfg::BM, BN = Cfg::BN, TI = Cfg::TI, TJ = Cfg::TJ, ST = Cfg::ST, PPW = Cfg::PPW, APW = Cfg::APW;
  constexpr int A_BYTES = Cfg::A_BYTES, STAGE = Cfg::STAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  combo_ts_begin(p.ts);
  const float* __restrict__ A = p.A;
  const float* __restrict__ Bm = p.B;
  const long long lda = p.lda, ldb = p.ldb;
  const int M = p.M, N = p.N, K = p.K;
  const ConvGeomF cg = p.cg;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int n_tiles = (N + BN - 1) / BN;
  const int tpb = ((M + BM - 1) / BM) * n_tiles;  // tiles per batch entry
  const int tiles = tpb * p.batch;
  const int G = gridDim.x;
  const int w = xcd_contiguous(blockIdx.x, G);  // every XCD owns a contiguous tile range: the n tiles of a token tile share an L2
  const int nst = K / kBK;

  // ---------------- issue cursor: (tile, stage) of the next stage to stream into the ring ----------------
  const int p_row = lane >> 2, p_chunk = lane & 3;
  int i_tile = w, i_s = 0, i_slot = 0, issued = 0;
  int i_tap = 0, i_cin0 = 0;
  unsigned tap_ok[APW];  // CONV: this lane's A rows -> 9-bit masks of the taps inside the map
#pragma unroll
  for (int u = 0; u < APW; ++u) tap_ok[u] = 0u;
  const float* pa[APW];
  const float* pb[PPW - APW];
  auto open_tile = [&]() {
    const int bi = i_tile / tpb, rem = i_tile - bi * tpb;
    const float* i_A = A + bi * p.sA;
    const float* i_B = Bm + bi * p.sB;
    const int i_m_blk = (rem / n_tiles) * BM;
    const int i_n_blk = (rem % n_tiles) * BN;
    i_s = 0; i_tap = 0; i_cin0 = 0;
#pragma unroll
    for (int u = 0; u < APW; ++u) {
      const int r = (wave + 4 * u) * 16 + p_row;
      const int c = p_chunk ^ ((r >> 2) & 3);  // swizzle on the SOURCE chunk, the LDS image stays lane-linear
      pa[u] = i_A + (long long)min(i_m_blk + r, M - 1) * lda + c * 4;
    }
#pragma unroll
    for (int u = 0; u < PPW - APW; ++u) {
      const int r = (wave + 4 * (u + APW) - Cfg::A_PIECES) * 16 + p_row;
      const int c = p_chunk ^ ((r >> 2) & 3);
      pb[u] = i_B + (long long)min(i_n_blk + r, N - 1) * ldb + c * 4;
    }
    if (CONV) {
#pragma unroll
      for (int u = 0; u < APW; ++u) {
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
    char* st = smem + i_slot * STAGE;
    const int k0 = i_s * kBK;
    const long long a_off = CONV ? (long long)((i_tap / 3 - 1) * cg.W + (i_tap % 3 - 1)) * lda + i_cin0 : (long long)k0;
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
      const int q = wave + 4 * u;  // wave-uniform piece (1 KiB = 16 rows x 64 B); q < A_PIECES: A rows, else B rows
      if (u < APW) {
        const float* src = pa[u < APW ? u : 0] + a_off;
        if (CONV) {
          const int c = p_chunk ^ (((q * 16 + p_row) >> 2) & 3);
          if (!((tap_ok[u < APW ? u : 0] >> i_tap) & 1u)) src = g_zero_row_f32 + c * 4;
        }
        glds16(src, st + q * 1024);
      } else {
        glds16(pb[u >= APW ? u - APW : 0] + k0, st + A_BYTES + (q - Cfg::A_PIECES) * 1024);
      }
    }
    ++issued;
    i_slot = i_slot == ST - 1 ? 0 : i_slot + 1;
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
#pragma unroll 1
  for (int q = 0; q < ST - 1; ++q) issue_next();

  // ---------------- LDS read addresses: lane (row m, k-half g) reads chunks 2g, 2g+1 of its rows ----------------
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int m = lane & 31, g = lane >> 5;
  const int sw = (m >> 2) & 3;  // a wave's rows are (multiple of 32) + m: they all share the swizzle of m
  const unsigned a_c0 = lds0 + (unsigned)((wm * TI * 32 + m) * 64 + ((2 * g) ^ sw) * 16);
  const unsigned a_c1 = lds0 + (unsigned)((wm * TI * 32 + m) * 64 + ((2 * g + 1) ^ sw) * 16);
  const unsigned b_c0 = lds0 + (unsigned)(A_BYTES + (wn * TJ * 32 + m) * 64 + ((2 * g) ^ sw) * 16);
  const unsigned b_c1 = lds0 + (unsigned)(A_BYTES + (wn * TJ * 32 + m) * 64 + ((2 * g + 1) ^ sw) * 16);

  int consumed = 0, c_slot = 0;
  for (int tile = w; tile < tiles; tile += G) {
    const int bi = tile / tpb, rem = tile - bi * tpb;
    const int m_blk = (rem / n_tiles) * BM, n_blk = (rem % n_tiles) * BN;
    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.C + bi * p.sC, 0, p.c_bytes, 0x00020000);
    // number of this wave's 32-row / 32-column sub-tiles that touch the matrix (wave-uniform)
    const int imax = min(TI, max(0, (M - m_blk - wm * TI * 32 + 31) / 32));
    const int jmax = min(TJ, max(0, (N - n_blk - wn * TJ * 32 + 31) / 32));
    const bool full = imax == TI && jmax == TJ;
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    for (int s = 0; s < nst; ++s) {
      // the stage to consume has landed once every OLDER vector-memory operation of this wave is done; the first stage of
      // a later tile also drains the epilogue stores of the previous tile (stores and loads share vmcnt)
      if (s == 0 && tile != w) wait_vm<0>();
      else wait_younger<PPW, ST>(issued - consumed - 1);
      __builtin_amdgcn_s_barrier();  // everybody's pieces landed; everybody finished reading the slot refilled below
      issue_next();
      const unsigned so = (unsigned)(c_slot * STAGE);
      f4v ra[TI][2], rb[TJ][2];
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        ra[i][0] = i == 0 ? lds_read128<0>(a_c0 + so) : lds_read128<2048>(a_c0 + so);
        ra[i][1] = i == 0 ? lds_read128<0>(a_c1 + so) : lds_read128<2048>(a_c1 + so);
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        rb[j][0] = j == 0 ? lds_read128<0>(b_c0 + so) : j == 1 ? lds_read128<2048>(b_c0 + so)
                 : j == 2 ? lds_read128<4096>(b_c0 + so) : lds_read128<6144>(b_c0 + so);
        rb[j][1] = j == 0 ? lds_read128<0>(b_c1 + so) : j == 1 ? lds_read128<2048>(b_c1 + so)
                 : j == 2 ? lds_read128<4096>(b_c1 + so) : lds_read128<6144>(b_c1 + so);
      }
      if constexpr (TI == 2 && TJ == 4) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(rb[0][0]), "+v"(rb[0][1]),
                       "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(rb[2][0]), "+v"(rb[2][1]), "+v"(rb[3][0]), "+v"(rb[3][1])
                     :
                     : "memory");
      } else if constexpr (TI == 2 && TJ == 2) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(rb[0][0]), "+v"(rb[0][1]),
                       "+v"(rb[1][0]), "+v"(rb[1][1])
                     :
                     : "memory");
      } else {
        static_assert((TI == 2 && TJ == 4) || (TI == 2 && TJ == 2) || (TI == 1 && TJ == 1), "add the register list of a new wave tile here");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(rb[0][0]), "+v"(rb[0][1]) : : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);  // (register-only MFMAs may not be hoisted above the inline-asm wait)
      if (full) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[i][t >> 2][t & 3], rb[j][t >> 2][t & 3], acc[i][j], 0, 0, 0);
      } else {  // edge tile: 32 x 32 sub-tiles that lie wholly outside the matrix are skipped (64 cycles each)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            if (i < imax && j < jmax) {
#pragma unroll
              for (int t = 0; t < 8; ++t)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[i][t >> 2][t & 3], rb[j][t >> 2][t & 3], acc[i][j], 0, 0, 0);
            }
      }
      ++consumed;
      c_slot = c_slot == ST - 1 ? 0 : c_slot + 1;
    }

    // epilogue: D tile = 32 tokens x 32 n; lane holds n = lane & 31 and tokens (e&3) + 8*(e>>2) + 4*(lane>>5).  Buffer
    // stores through one descriptor over C: the hardware range check drops the rows >= M of the last token tile.
    auto epilogue = [&](auto relu_tag) {
      constexpr bool RELU = decltype(relu_tag)::value;
      const unsigned uld = (unsigned)p.ldc * 4u, uld5 = 5u * uld;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int n = n_blk + (wn * TJ + j) * 32 + m;
        if (n >= N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
        unsigned off = ((unsigned)(m_blk + wm * TI * 32 + 4 * g) * (unsigned)p.ldc + (unsigned)n) * 4u;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float v = acc[i][j][e] + bv;
            if (RELU) v = fmaxf(v, 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), c_rsrc, off, 0, 0);
            off += (e & 3) == 3 ? uld5 : uld;  // rows 0-3, 8-11, 16-19, 24-27 (+4g); the next i starts 32 rows on
          }
      }
    };
    if (p.relu) epilogue(std::true_type{});
    else epilogue(std::false_type{});
  }
  combo_ts_end(p.ts);
}

int n_cu_cached() {
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return cus > 0 ? cus : 256;
  }();
  return n_cu;
}

template <bool CONV, typename Cfg>
int launch_cfg(F32Args a, hipStream_t stream) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_f32_kernel<CONV, Cfg>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const long long tiles = ((a.M + Cfg::BM - 1LL) / Cfg::BM) * ((a.N + Cfg::BN - 1LL) / Cfg::BN) * a.batch;
  if (tiles > 0x7fffffffLL) return COMBO_EINVAL;
  const long long slots = (long long)Cfg::WGS * n_cu_cached();
  const int grid = (int)(tiles < slots ? tiles : slots);
  a.ts = combo_timing_next_slot(COMBO_TS_GEMM_F32, 2.0 * a.M * a.N * a.K * a.batch);
  hipLaunchKernelGGL((gemm_nt_f32_kernel<CONV, Cfg>), dim3((unsigned)grid), dim3(256), Cfg::LDS, stream, a);
  return (int)hipGetLastError();
}

// Tile choice.  The kernel is MFMA-bound, so its time is the load of the busiest CU: tiles spread round-robin over the CUs
// (persistent workgroups, XCD-contiguous), each costing BM*BN*K MACs; the smallest "ceil(tiles / CUs) * BM * BN" wins, ties
// go to the larger tile (fewer operand bytes per flop: 64 x 64 tiles pull 16 flop/B from L2, enough only for small problems).
template <bool CONV>
int launch_f32(F32Args a, hipStream_t stream, int force) {
  static const int env_force = [] { const char* e = getenv("COMBO_F32_TILE"); return e ? atoi(e) : 0; }();  // 1 wide, 2 mid, 3 skinny (A/B)
  if (!force) force = env_force;
  const long long cus = n_cu_cached();
  auto load = [&](int bm, int bn) {
    const long long t = ((a.M + bm - 1LL) / bm) * ((a.N + bn - 1LL) / bn) * a.batch;
    return ((t + cus - 1) / cus) * bm * bn;
  };
  int pick = force;
  if (!pick) {
    const long long lw = load(256, 128), lm = load(128, 128), ls = load(64, 64);
    pick = 1;
    long long best = lw;
    if (lm < best) { best = lm; pick = 2; }
    const double flops = 2.0 * a.M * a.N * a.K * a.batch;
    if (ls < best && (flops < 3.0e9 || ls * 4 <= best * 3)) pick = 3;  // skinny: small problems, or >= 25 % less load
  }
  if (pick == 3) return launch_cfg<CONV, FSkinny>(a, stream);
  if (pick == 2) return launch_cfg<CONV, FMid>(a, stream);
  return launch_cfg<CONV, FWide>(a, stream);
}

bool args_ok(const float* A, long long lda, const float* B, long long ldb, const float* C, long long ldc, long long M, int N, int K,
             int batch) {
  return A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && K % kBK == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
         !((uintptr_t)A & 15) && !((uintptr_t)B & 15) && M <= 0x7fffffffLL && ((M - 1) * ldc + N) * 4 < 0x7fffffffLL;
}

}  // namespace

extern "C" int combo_gemm_nt_f32(const float* A, long long lda, const float* B, long long ldb, const float* bias, float* C,
                                 long long ldc, int M, int N, int K, int relu, combo_stream_t stream) {
  if (!args_ok(A, lda, B, ldb, C, ldc, M, N, K, 1)) return COMBO_EINVAL;
  F32Args a{A, lda, B, ldb, bias, C, ldc, M, N, K, relu, (int)(((M - 1LL) * ldc + N) * 4), 1, 0, 0, 0, ConvGeomF{1, 1, K}, nullptr};
  return launch_f32<false>(a, (hipStream_t)stream, 0);
}

extern "C" int combo_gemm_nt_batched_f32(const float* A, long long lda, long long sA, const float* B, long long ldb, long long sB,
                                         float* C, long long ldc, long long sC, int M, int N, int K, int batch, int relu,
                                         combo_stream_t stream) {
  if (!args_ok(A, lda, B, ldb, C, ldc, M, N, K, batch) || sA % 4 != 0 || sB % 4 != 0) return COMBO_EINVAL;
  F32Args a{A, lda, B, ldb, nullptr, C, ldc, M, N, K, relu, (int)(((M - 1LL) * ldc + N) * 4), batch, sA, sB, sC,
            ConvGeomF{1, 1, K}, nullptr};
  return launch_f32<false>(a, (hipStream_t)stream, 0);
}

extern "C" int combo_conv3x3_nhwc_f32(const float* X, long long ldx, const float* Wm, const float* bias, float* Y, long long ldy,
                                      int B, int H, int W, int Cin, int Cout, int relu, combo_stream_t stream) {
  const long long M = (long long)B * H * W;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % kBK != 0 || M > 0x7fffffffLL / 4 ||
      !args_ok(X, ldx, Wm, 9LL * Cin, Y, ldy, M, Cout, 9 * Cin, 1))
    return COMBO_EINVAL;
  F32Args a{X, ldx, Wm, 9LL * Cin, bias, Y, ldy, (int)M, Cout, 9 * Cin, relu, (int)(((M - 1) * ldy + Cout) * 4), 1, 0, 0, 0,
            ConvGeomF{H, W, Cin}, nullptr};
  return launch_f32<true>(a, (hipStream_t)stream, 0);
}
