// gemm_nt3: C[M,N] = A[M,K] . B[N,K]^T (+ bias) (+ ReLU) (masked), fp32 accuracy from 3 bf16 MFMA products (hi.hi + lo.hi +
// hi.lo, rounded `hi`: ~2^-17 relative per product) - the input-gradient GEMM dX = dY . W of every dense layer, the implicit-GEMM
// 3x3 convolution's input gradient, the batched mask-logit gradients, and (products = 1) the head's bf16 forward mode.
//
// Why v3 (round 4).  Its predecessor gemm_nt2 (round 3: two 4-wave workgroups per CU, a ring of 3 stages) did not overlap its
// phases.  Ablation of that kernel on MI355X (41160 x 1024 -> 256, hot loop): 93 us = 25 (loop skeleton: waits, cursor, address
// arithmetic) + 30 (MFMA, i.e. the matrix pipe at ~2.2 PF while it runs) + 31 (LDS-DMA issue + landing) + 8 (split) + 3 (LDS reads)
// + 3 (stores): the SUM of the parts - equal tiles keep the two workgroups of a CU in phase.  Inside the training step it ran at
// 0.18 of its 833 TFLOP/s ceiling and 0.22 of HBM, and the decoder / res4 / res5 layers (few output tiles, long K) used 1/8 of the
// chip.  v3 is gemm_f32.hip's skeleton - ONE persistent workgroup per CU, LDS-DMA ring of 6 stages with counted vmcnt across a raw
// s_barrier, every wave software-pipelined - re-timed for a matrix instruction that is 16x faster than the f32 one:
//   * a stage (BK = 16) is ONE 32x32x16 MFMA triple per 32 x 32 sub-tile; the wide tile (256 x 128) is worked by 8 waves (two per
//     SIMD, 32 x 128 each: 12 MFMAs = 384 pipe cycles per wave and stage): while one wave of a SIMD sits in an LDS-DMA issue or an
//     LDS wait the other one feeds the pipe;
//   * phase 0 (hi.hi and lo.hi products): the ds_reads of THIS stage's B `lo` groups and of the NEXT stage's raw A rows go out
//     first, the DMA pieces of the stage entering the ring are issued between the MFMAs;
//   * phase 1 (hi.lo products): the next stage's B `hi` groups are read into the registers phase 0 just freed and the next
//     stage's A rows are split (hi = rne_bf16(x), lo = rne_bf16(x - hi): 6 VALU per pair) between the MFMAs, into the other half
//     of a two-set register ping-pong (no copies);
//   * one barrier per stage, at its top: "stage s + 1 has landed for everybody and everybody is done with slot s - 1" (the slot
//     the DMA of this stage refills), with ST - 3 younger stages left in flight;
//   * NO branch in the stream: past the last tile the cursor goes on issuing dummy stages (the zero row), so exactly ST - 3
//     younger stages are in flight at every wait and prefetching "the next stage" is always legal; ablation bits and the product
//     count are template parameters (a run-time test per MFMA cost more than the MFMA);
//   * the epilogue's dwordx4 stores are NOT waited for: the vmcnt budget of the next tile's first ST - 2 stages is raised by the
//     number of stores (they are younger than every ring load those stages need), so the ring never drains at a tile boundary;
//     operands swapped (D = W . X^T) as in gemm_f32: a lane owns 4 consecutive n of one token;
//   * few output tiles and a long K (4000 x 2048 -> 256, the backbones' res4 / res5 layers): split-K as batch entries of one
//     launch + a fixed-order finishing sum (combo_gemm_nt_x3_splitk_*): 63 -> 27 us.
// Measured (tools/bench_nt3.py): one round of wide tiles at K = 1024 (32768 x 1024 -> 256) 55 us = 0.93 PF of bf16 MFMA issue
// (hipBLASLt's plain bf16 GEMM sustains 1.1 - 1.4 PF under the 1400 W cap); without DMA 37 us, without MFMA 45 us, i.e. the
// phases overlap now; the training step's input-gradient GEMMs 7.17 -> 6.4 ms.  What is left is tile quantisation (41160 rows
// x 256 columns are 1.26 rounds of wide tiles) and the 66 decoder launches of 0.5 GFLOP at their ~8 us latency floor.
// Tile shapes / row split: as gemm_f32.hip (wide 256 x 128, mid 128 x 128, skinny 64 x 64; whole rounds of large tiles + the
// remaining rows on small ones).
#include <cstdlib>
#include <type_traits>

#include "combo_common.h"
#include "gemm_nt3.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) unsigned u4v;

__device__ __forceinline__ unsigned pack_rne(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// fp16 pieces (round 6: the fp32-GRADE forward mode).  hi = rne_f16(x), lo = rne_f16(x - hi): 22 mantissa bits, the dropped lo.lo
// term and the rounding of lo are ~2^-22 |x.w| (bf16 pieces: 2^-16 / 2^-17) - at the price of fp16's RANGE: |x| < 65 504, and the
// pieces of small operands are fp16 subnormals (absolute floor 2^-25; v_cvt_pk_f16_f32 and v_mfma_*_f16 keep subnormals on gfx950:
// tools/ubench/mfma_f16_denorm.hip).  Forward activations / weights of the normalised head only - gradients stay on bf16 pieces.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
__device__ __forceinline__ unsigned pack_rne_f16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <bool F16>
__device__ __forceinline__ unsigned pack2(float a, float b) { return F16 ? pack_rne_f16(a, b) : pack_rne(a, b); }
template <bool F16>
__device__ __forceinline__ void unpack2(unsigned h, float& f0, float& f1) {
  if constexpr (F16) {
    const f16x2 v = __builtin_bit_cast(f16x2, h);
    f0 = (float)v[0];
    f1 = (float)v[1];
  } else {
    f0 = __uint_as_float(h << 16);
    f1 = __uint_as_float(h & 0xffff0000u);
  }
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int OFF>
__device__ __forceinline__ f4v lds_read128(unsigned addr) {  // inline asm: hipcc drains the LDS-DMA queue before a visible ds_read
  f4v r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// s_waitcnt vmcnt(n) for a run-time n (the immediate is a 6-bit field): rounds n DOWN to one of a few values - waiting for more
// than necessary is always correct
__device__ __forceinline__ void wait_vm_upto(int n) {
  if (n >= 48) wait_vm<48>();
  else if (n >= 36) wait_vm<36>();
  else if (n >= 30) wait_vm<30>();
  else if (n >= 24) wait_vm<24>();
  else if (n >= 20) wait_vm<20>();
  else if (n >= 16) wait_vm<16>();
  else if (n >= 15) wait_vm<15>();
  else if (n >= 12) wait_vm<12>();
  else if (n >= 10) wait_vm<10>();
  else if (n >= 9) wait_vm<9>();
  else if (n >= 8) wait_vm<8>();
  else if (n >= 6) wait_vm<6>();
  else if (n >= 5) wait_vm<5>();
  else if (n >= 4) wait_vm<4>();
  else if (n >= 3) wait_vm<3>();
  else if (n >= 2) wait_vm<2>();
  else wait_vm<0>();
}
__device__ __forceinline__ void glds16(const float* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

constexpr int kBK = 16;
constexpr float kF16Unscale = 1.0f / (float)(1 << COMBO_F16_BSCALE_LOG2);
constexpr int kMaxBiasN = 2048;  // the bias vector lives in LDS (ordinary global loads next to the LDS-DMA stream drain the ring)

__device__ __attribute__((aligned(64))) float g_zero_row3[16];  // zero-initialised: the source of padded conv taps

template <int WM_, int WN_, int TI_, int TJ_, int ST_>
struct N3Cfg {
  static constexpr int WM = WM_, WN = WN_, TI = TI_, TJ = TJ_, ST = ST_;
  static constexpr int NW = WM * WN;  // waves per workgroup: 4 (one per SIMD) or 8 (two per SIMD: one computes while the other
                                      // sits in an LDS-DMA issue or an LDS wait)
  static constexpr int BM = WM * TI * 32, BN = WN * TJ * 32;
  static constexpr int A_BYTES = BM * kBK * 4, B_BYTES = BN * kBK * 4, STAGE = A_BYTES + B_BYTES;
  static constexpr int A_PIECES = A_BYTES / 1024, PIECES = STAGE / 1024, PPW = PIECES / NW, APW = A_PIECES / NW;
  static constexpr int RING = ST * STAGE;
  static constexpr int LDS = RING + kMaxBiasN * 4;
  static constexpr int STORES = TI * TJ * 4;  // dwordx4 stores per lane and tile
  static_assert((NW == 4 || NW == 8) && PIECES % NW == 0 && A_PIECES % NW == 0, "the waves share the DMA pieces evenly");
  static_assert(ST >= 4 && (ST - 3) * PPW + STORES <= 63, "vmcnt is a 6-bit counter");
  static_assert(LDS <= 160 * 1024, "LDS budget of a CU");
};
typedef N3Cfg<8, 1, 1, 4, 6> NWide;     // 8 waves x (32 x 128): 24 KiB stages, 144 KiB ring
typedef N3Cfg<2, 2, 2, 2, 8> NMid;      // 16 KiB stages, 128 KiB ring
typedef N3Cfg<2, 2, 1, 1, 16> NSkinny;  //  8 KiB stages, 128 KiB ring = a whole K = 256 panel in flight
typedef N3Cfg<4, 1, 2, 2, 7> NTall;     // 4 waves x (64 x 64) stacked: 256 x 64 tiles for 64-wide outputs (res2's 64-channel layers)

struct N3Args {
  const float* A; long long lda;
  const float* B; long long ldb;   // pre-split image (combo_presplit_bf16x2_*): per 8 k a 16-B bf16 hi group + a 16-B lo group
  const float* bias;               // [N] or nullptr
  const float* mask;               // [M, N] (pitch ldc, batch stride sC) or nullptr: C = mask > 0 ? value : 0, applied last
  const float* add;                // [M, N] (same pitch / stride) or nullptr: C += add before the ReLU (the residual branch of a
                                   // bottleneck block; the other gradient arriving at a block input)
  float* C; long long ldc;
  int M, N, K, relu, c_bytes, batch, vec_store, dbg, products;
  int f16;                         // 1: fp16 pieces (the image was made by the fp16 pre-split), 0: bf16
  float unscale;                   // fp16 pieces: the epilogue's factor (2^-8 for a weight image split from 2^8 . w, else 1)
  long long sA, sB, sC;
  combo_nt3_conv cg;
  unsigned long long* ts;
  unsigned long long* prof;         // ABL bit 128: per (workgroup, wave) cycle sums of the stage's three segments (tools/prof_nt3_stage.py)
};

// P3: three products (fp32-accurate) or one (plain bf16); ABL: ablation bits, COMPILE-TIME (a run-time test per bit costs a scalar
// branch per use, ~1 000 cycles per stage in total - more than the stage itself): 1 no DMA, 2 no LDS reads, 4 no barrier, 8 no
// stores, 16 no split, 32 no MFMA, 128 (not an ablation: per-segment cycle stamps of every stage); the instances listed in launch_cfg3 exist (COMBO_NT3_DBG, tools/bench_nt3.py)
template <bool CONV, typename Cfg, bool P3, int ABL, bool AUX, bool F16 = false>
__global__ void __launch_bounds__(Cfg::NW * 64, Cfg::NW / 4)
gemm_nt3_kernel(const N3Args p) {
  constexpr int NW = Cfg::NW;
  constexpr int dbg = ABL;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, TI = Cfg::TI, TJ = Cfg::TJ, ST = Cfg::ST, PPW = Cfg::PPW, APW = Cfg::APW;
  constexpr int A_BYTES = Cfg::A_BYTES, STAGE = Cfg::STAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  combo_ts_begin(p.ts);
  const float* __restrict__ A = p.A;
  const float* __restrict__ Bm = p.B;
  const long long lda = p.lda, ldb = p.ldb;
  const int M = p.M, N = p.N, K = p.K;
  const combo_nt3_conv cg = p.cg;
  const int cs = CONV && cg.stride == 2 ? 2 : 1;                          // CONV: the output map (M = batch x ho x wo tokens)
  const int ho = (cg.H + cs - 1) / cs, wo = (cg.W + cs - 1) / cs;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int n_tiles = (N + BN - 1) / BN;
  const int tpb = ((M + BM - 1) / BM) * n_tiles;  // tiles per batch entry
  const int tiles = tpb * p.batch;
  const int G = gridDim.x;
  const int w = xcd_contiguous(blockIdx.x, G);  // every XCD owns a contiguous tile range: the n tiles of a token tile share an L2
  const int nst = K / kBK;
  if (w >= tiles) {  // (grid <= tiles by construction; kept for safety: no workgroup may skip the barriers below)
    combo_ts_end(p.ts);
    return;
  }

  // the bias vector goes to LDS once (zero beyond N), before any LDS-DMA is in flight
  const unsigned bias_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)Cfg::RING;
  const int n_pad = min(kMaxBiasN, (N + 3) & ~3);
  if (p.bias) {
    float* bl = reinterpret_cast<float*>(smem + Cfg::RING);
    for (int n = threadIdx.x; n < n_pad; n += NW * 64) bl[n] = n < N ? p.bias[n] : 0.f;
    __syncthreads();
  }

  // ---------------- issue cursor: (tile, stage) of the next stage to stream into the ring ----------------
  const int p_row = lane >> 2, p_chunk = lane & 3;
  int i_tile = w, i_s = 0, i_slot = 0, issued = 0;
  int i_tap = 0, i_cin0 = 0;
  unsigned tap_ok[APW];  // CONV: this lane's A rows -> 9-bit masks of the taps inside the map
#pragma unroll
  for (int u = 0; u < APW; ++u) tap_ok[u] = 0u;
  const float* pa[APW];
  const float* pb[PPW - APW];
  int k_mul = 1;  // 0 once the cursor has run past the last tile: the stream goes on with dummy stages (the zero row, k offset 0)
  // so that EXACTLY ST - 3 younger stages are in flight at every wait and no piece needs a branch; their slots are free by
  // construction and nobody reads them
  auto open_tile = [&]() __attribute__((always_inline)) {
    if (i_tile >= tiles) {
      k_mul = 0;
#pragma unroll
      for (int u = 0; u < APW; ++u) { pa[u] = g_zero_row3 + p_chunk * 4; tap_ok[u] = 0x1ffu; }
#pragma unroll
      for (int u = 0; u < PPW - APW; ++u) pb[u] = g_zero_row3 + p_chunk * 4;
      i_s = 0; i_tap = 0; i_cin0 = 0;
      return;
    }
    const int bi = i_tile / tpb, rem = i_tile - bi * tpb;
    const float* i_A = A + bi * p.sA;
    const float* i_B = Bm + bi * p.sB;
    const int i_m_blk = (rem / n_tiles) * BM;
    const int i_n_blk = (rem % n_tiles) * BN;
    i_s = 0; i_tap = CONV ? bi * cg.tap_step : 0; i_cin0 = 0;  // (split 3x3 convolution: batch entry bi owns taps bi * tap_step ...)
#pragma unroll
    for (int u = 0; u < APW; ++u) {
      const int r = (wave + NW * u) * 16 + p_row;
      const int c = p_chunk ^ ((r >> 2) & 3);  // swizzle on the SOURCE chunk, the LDS image stays lane-linear
      long long row = min(i_m_blk + r, M - 1);
      if (CONV && cg.stride == 2) {  // output token -> the input token under its centre tap
        const int t = (int)row, xo = t % wo, yo = (t / wo) % ho, b = t / (wo * ho);
        row = ((long long)b * cg.H + 2 * yo) * cg.W + 2 * xo;
      }
      pa[u] = i_A + row * lda + c * 4;
    }
#pragma unroll
    for (int u = 0; u < PPW - APW; ++u) {
      const int r = (wave + NW * (u + APW) - Cfg::A_PIECES) * 16 + p_row;
      const int c = p_chunk ^ ((r >> 2) & 3);
      pb[u] = i_B + (long long)min(i_n_blk + r, N - 1) * ldb + c * 4;
    }
    if (CONV) {
#pragma unroll
      for (int u = 0; u < APW; ++u) {
        const int t = min(i_m_blk + (wave + NW * u) * 16 + p_row, M - 1);
        const int x = (t % wo) * cs, y = ((t / wo) % ho) * cs;  // centre tap in the input map
        unsigned ok = 0u;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int yy = y + tap / 3 - cg.pad, xx = x + tap % 3 - cg.pad;
          if (yy >= 0 && yy < cg.H && xx >= 0 && xx < cg.W) ok |= 1u << tap;
        }
        tap_ok[u] = ok;
      }
    }
  };
  // one DMA piece (1 KiB = 16 rows x 64 B, wave-uniform destination) of the stage under the issue cursor
  auto issue_piece = [&](auto u_tag) __attribute__((always_inline)) {
    constexpr int u = decltype(u_tag)::value;
    if (dbg & 1) return;
    char* st = smem + i_slot * STAGE;
    const int q = wave + NW * u;  // q < A_PIECES: A rows, else B rows
    if constexpr (u < APW) {
      const long long a_off = (CONV ? (long long)((i_tap / 3 - cg.pad) * cg.W + (i_tap % 3 - cg.pad)) * lda + i_cin0 : (long long)(i_s * kBK)) * k_mul;
      const float* src = pa[u] + a_off;
      if (CONV) {
        const int c = p_chunk ^ (((q * 16 + p_row) >> 2) & 3);
        if (!((tap_ok[u] >> i_tap) & 1u)) src = g_zero_row3 + c * 4;
      }
      glds16(src, st + q * 1024);
    } else {
      glds16(pb[u - APW] + i_s * kBK * k_mul, st + A_BYTES + (q - Cfg::A_PIECES) * 1024);
    }
  };
  auto issue_finish = [&]() __attribute__((always_inline)) {  // advance the cursor past the stage just issued
    ++issued;
    i_slot = i_slot == ST - 1 ? 0 : i_slot + 1;
    ++i_s;
    if (CONV) {
      i_cin0 += kBK;
      if (i_cin0 == cg.Cin) { i_cin0 = 0; ++i_tap; }
    }
    if (i_s == nst) {
      if (i_tile < tiles) i_tile += G;
      open_tile();
    }
  };
  auto issue_u = [&](int u) __attribute__((always_inline)) {  // u is a compile-time constant after unrolling
    if (u == 0) issue_piece(std::integral_constant<int, 0>{});
    if (u == 1 && PPW > 1) issue_piece(std::integral_constant<int, (PPW > 1 ? 1 : 0)>{});
    if (u == 2 && PPW > 2) issue_piece(std::integral_constant<int, (PPW > 2 ? 2 : 0)>{});
    if (u == 3 && PPW > 3) issue_piece(std::integral_constant<int, (PPW > 3 ? 3 : 0)>{});
    if (u == 4 && PPW > 4) issue_piece(std::integral_constant<int, (PPW > 4 ? 4 : 0)>{});
    if (u == 5 && PPW > 5) issue_piece(std::integral_constant<int, (PPW > 5 ? 5 : 0)>{});
    static_assert(PPW <= 6, "add the pieces of a larger stage here");
    if (u == PPW) issue_finish();
  };
  open_tile();
#pragma unroll 1
  for (int q = 0; q < ST - 1; ++q) {
#pragma unroll
    for (int u = 0; u <= PPW; ++u) issue_u(u);
  }

  // ---------------- LDS read addresses: lane (row r, k-half g) reads chunks 2g, 2g+1 of its rows ----------------
  // A rows (fp32): chunk 2g = k 8g..8g+3, chunk 2g+1 = k 8g+4..8g+7.  B rows (pre-split image): chunk 2g = bf16 hi of k 8g..8g+7,
  // chunk 2g+1 = bf16 lo of the same k.
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int r = lane & 31, g = lane >> 5;
  const int sw = (r >> 2) & 3;  // a wave's rows are (multiple of 32) + r: they all share the swizzle of r
  const unsigned a_c0 = lds0 + (unsigned)((wm * TI * 32 + r) * 64 + ((2 * g) ^ sw) * 16);
  const unsigned a_c1 = lds0 + (unsigned)((wm * TI * 32 + r) * 64 + ((2 * g + 1) ^ sw) * 16);
  const unsigned b_hi = lds0 + (unsigned)(A_BYTES + (wn * TJ * 32 + r) * 64 + ((2 * g) ^ sw) * 16);
  const unsigned b_lo = lds0 + (unsigned)(A_BYTES + (wn * TJ * 32 + r) * 64 + ((2 * g + 1) ^ sw) * 16);

  // operand registers: two sets (ping-pong over the stage parity) of split A fragments and B hi groups; one set of B lo groups
  // and of raw A rows (both live for less than a stage)
  bf16x8 ah[2][TI], al[2][TI], bh[2][TJ], bl[TJ];
  f4v ra[TI][2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
    for (int i = 0; i < TI; ++i) { ah[s2][i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; al[s2][i] = ah[s2][i]; }
#pragma unroll
    for (int j = 0; j < TJ; ++j) bh[s2][j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
#pragma unroll
  for (int j = 0; j < TJ; ++j) bl[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < TI; ++i) ra[i][0] = ra[i][1] = f4v{1.f, 1.f, 1.f, 1.f};

  auto read_b = [&](bf16x8 (&dst)[TJ], unsigned base) __attribute__((always_inline)) {
    if (dbg & 2) return;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const f4v v = j == 0 ? lds_read128<0>(base) : j == 1 ? lds_read128<2048>(base) : j == 2 ? lds_read128<4096>(base) : lds_read128<6144>(base);
      dst[j] = __builtin_bit_cast(bf16x8, v);
    }
  };
  auto read_a = [&](unsigned so) __attribute__((always_inline)) {
    if (dbg & 2) return;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      ra[i][0] = i == 0 ? lds_read128<0>(a_c0 + so) : lds_read128<2048>(a_c0 + so);
      ra[i][1] = i == 0 ? lds_read128<0>(a_c1 + so) : lds_read128<2048>(a_c1 + so);
    }
  };
  // s_waitcnt lgkmcnt(0) tied to the registers the pending ds_reads write (hipcc may not move their uses above it)
  auto wait_b = [&](bf16x8 (&x)[TJ]) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TJ == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : : "memory");
    else if constexpr (TJ == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]) : : "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]) : : "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto wait_ab = [&](bf16x8 (&x)[TJ]) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TI == 2 && TJ == 4)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]) : : "memory");
    else if constexpr (TI == 2 && TJ == 2)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]) : : "memory");
    else if constexpr (TI == 1 && TJ == 4)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(ra[0][0]), "+v"(ra[0][1]) : : "memory");
    else {
      static_assert((TI == 2 && TJ == 4) || (TI == 2 && TJ == 2) || (TI == 1 && TJ == 4) || (TI == 1 && TJ == 1), "add the register list of a new wave tile here");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(ra[0][0]), "+v"(ra[0][1]) : : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // one quarter of the split of row-tile i (two floats -> one packed bf16 pair of hi and of lo); q = 0..3
  unsigned hw_[TI][4], lw_[TI][4];
  auto split_q = [&](int i, int q) __attribute__((always_inline)) {
    const float v0 = q < 2 ? ra[i][0][2 * q] : ra[i][1][2 * (q - 2)];
    const float v1 = q < 2 ? ra[i][0][2 * q + 1] : ra[i][1][2 * (q - 2) + 1];
    if (dbg & 16) { hw_[i][q] = __float_as_uint(v0); lw_[i][q] = __float_as_uint(v1); return; }
    const unsigned h = pack2<F16>(v0, v1);  // hi = rne_bf16(x): the dropped lo.lo term is <= 2^-16 |x.w| and unbiased
    hw_[i][q] = h;
    float h0, h1;
    unpack2<F16>(h, h0, h1);
    lw_[i][q] = pack2<F16>(v0 - h0, v1 - h1);
  };
  auto split_commit = [&](bf16x8 (&dh)[TI], bf16x8 (&dl)[TI]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      dh[i] = __builtin_bit_cast(bf16x8, u4v{hw_[i][0], hw_[i][1], hw_[i][2], hw_[i][3]});
      dl[i] = __builtin_bit_cast(bf16x8, u4v{lw_[i][0], lw_[i][1], lw_[i][2], lw_[i][3]});
    }
  };

  f32x16 acc[TI][TJ];
  f32x16 zero16;
#pragma unroll
  for (int q = 0; q < 16; ++q) zero16[q] = 0.f;

  // ---------------- epilogue of one tile: lane (token r, half g) owns n = 8q + 4g + (0..3), q = 0..3, of every sub-tile ----------------
  struct EpiCtx { int m_blk, n_blk, bi; };
  // Early auxiliary operand (round 5).  The epilogue's `add` / `mask` loads used to be issued in the epilogue, i.e. BEHIND the
  // ST - 1 stages of the next tile already streaming into the ring (120 KiB per CU for the wide tile): the wait for them was a
  // wait for the whole ring - one pipeline refill per tile, 5 us against 2.6 us of matrix work at K = 256 (the FFN's masked dX GEMM ran
  // at 2.2 x its HBM time).  Now ONE of the two operands (add if present, else mask) is requested ST - 1 stages before the tile
  // ends, in front of the next tile's first stage: by the epilogue it has landed, and the counted waits in between allow its
  // AUXN loads to stay outstanding (they are younger than every ring load those stages need).  Costs TI * TJ * 16 registers.
  // (not on the 8-wave convolution instance: its 256-register budget has no room for them - 60 spilled registers - and it keeps
  //  the in-epilogue loads)
  // AUX: an instance of its own (the request's addressing costs the plain launches ~40 spilled scalar registers otherwise)
  constexpr bool AUX_OK = AUX && !(CONV && NW == 8);
  constexpr int AUXN = AUX_OK ? TI * TJ * 4 : 0;
  u4v aux[AUX_OK ? TI : 1][AUX_OK ? TJ : 1][4];
  const bool aux_is_add = p.add != nullptr;
  const bool aux_early = AUX_OK && p.vec_store && (p.add || p.mask) && !(dbg & 8);
  auto aux_request = [&](const EpiCtx& ec) __attribute__((always_inline)) {
    if constexpr (!AUX_OK) return;
    const float* src = aux_is_add ? p.add : p.mask;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src + ec.bi * p.sC), 0, p.c_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int row = ec.m_blk + (wm * TI + i) * 32 + r;
        const int nb = ec.n_blk + (wn * TJ + j) * 32 + 4 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n0 = nb + 8 * q;
          const unsigned off = n0 < N ? ((unsigned)row * (unsigned)p.ldc + (unsigned)n0) * 4u : 0xfffffff0u;
          aux[AUX_OK ? i : 0][AUX_OK ? j : 0][q] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, off, 0, 0);
        }
      }
  };
  auto epilogue = [&](const EpiCtx& ec) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.C + ec.bi * p.sC, 0, p.c_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t m_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.mask ? p.mask + ec.bi * p.sC : p.C), 0, p.c_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t d_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.add ? p.add + ec.bi * p.sC : p.C), 0, p.c_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        __builtin_amdgcn_sched_barrier(0);  // one sub-tile at a time: 16 accumulator registers in flight, not 128
        const int row = ec.m_blk + (wm * TI + i) * 32 + r;
        const int nb = ec.n_blk + (wn * TJ + j) * 32 + 4 * g;
        if (dbg & 8) {  // keep the accumulators alive without storing them
#pragma unroll
          for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(acc[i][j][e]));
          continue;
        }
        if (p.vec_store) {
          f4v bv[4];
          if (p.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = lds_read128<0>(bias_lds + (unsigned)min(nb + 8 * q, n_pad - 4) * 4u);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]) : : "memory");
          }
          // add / mask: one register set, used twice (the wide tiles have no registers to spare).  These loads are the youngest
          // vector-memory operations: waiting for them drains the ring - such launches pay one pipeline refill per tile.
          f4v v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v[q] = f4v{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            if constexpr (F16) v[q] *= p.unscale;  // (a weight image holds 2^8 . w: gemm_nt3.h)
            if (p.bias) v[q] += bv[q];
          }
          u4v mv[4];
          if (p.add) {
            if (AUX_OK && aux_early) {  // (requested ST - 1 stages ago: aux_request)
#pragma unroll
              for (int q = 0; q < 4; ++q) mv[q] = aux[AUX_OK ? i : 0][AUX_OK ? j : 0][q];
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int n0 = nb + 8 * q;
                const unsigned off = n0 < N ? ((unsigned)row * (unsigned)p.ldc + (unsigned)n0) * 4u : 0xfffffff0u;
                mv[q] = __builtin_amdgcn_raw_buffer_load_b128(d_rsrc, off, 0, 0);
              }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += __builtin_bit_cast(f4v, mv[q]);
          }
          if (p.relu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              v[q].x = fmaxf(v[q].x, 0.f); v[q].y = fmaxf(v[q].y, 0.f); v[q].z = fmaxf(v[q].z, 0.f); v[q].w = fmaxf(v[q].w, 0.f);
            }
          }
          if (p.mask) {
            __builtin_amdgcn_sched_barrier(0);  // (the mask loads reuse the registers of the add loads: keep them behind)
            if (AUX_OK && aux_early && !aux_is_add) {
#pragma unroll
              for (int q = 0; q < 4; ++q) mv[q] = aux[AUX_OK ? i : 0][AUX_OK ? j : 0][q];
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int n0 = nb + 8 * q;
                const unsigned off = n0 < N ? ((unsigned)row * (unsigned)p.ldc + (unsigned)n0) * 4u : 0xfffffff0u;
                mv[q] = __builtin_amdgcn_raw_buffer_load_b128(m_rsrc, off, 0, 0);
              }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              v[q].x = __uint_as_float(mv[q].x) > 0.f ? v[q].x : 0.f; v[q].y = __uint_as_float(mv[q].y) > 0.f ? v[q].y : 0.f;
              v[q].z = __uint_as_float(mv[q].z) > 0.f ? v[q].z : 0.f; v[q].w = __uint_as_float(mv[q].w) > 0.f ? v[q].w : 0.f;
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n0 = nb + 8 * q;
            // rows >= M fall outside the descriptor's range and are dropped; columns >= N are steered there as well
            const unsigned off = n0 < N ? ((unsigned)row * (unsigned)p.ldc + (unsigned)n0) * 4u : 0xfffffff0u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v[q]), c_rsrc, off, 0, 0);
          }
        } else {  // N or ldc not a multiple of 4: scalar stores
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int n = nb + 8 * (e >> 2) + (e & 3);
            float v = acc[i][j][e];
            if constexpr (F16) v *= p.unscale;
            if (p.bias) {
              float b;
              asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(bias_lds + (unsigned)min(n, n_pad - 1) * 4u) : "memory");
              v += b;
            }
            const unsigned off = n < N ? ((unsigned)row * (unsigned)p.ldc + (unsigned)n) * 4u : 0xfffffff0u;
            if (p.add) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(d_rsrc, off, 0, 0));
            if (p.relu) v = fmaxf(v, 0.f);
            if (p.mask) v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(m_rsrc, off, 0, 0)) > 0.f ? v : 0.f;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), c_rsrc, off, 0, 0);
          }
        }
      }
  };

  // ---------------- one stage ----------------
  // PAR: which register set holds this stage's operands (the next stage's go to set PAR ^ 1).  first: first stage of a tile (the
  // hi.hi products start the accumulators from zero).  `relaxed`: the epilogue stores of the previous tile may still be in
  // flight behind the ring loads we wait for.
  // ABL bit 128 (tools/prof_nt3_stage.py): s_memtime at three points of every stage - top (before the counted wait), behind the barrier, behind
  // phase 0's wait; a mark CONSUMES the time stamp issued at the previous mark (long landed: no wait) and issues its own, so the stamps never
  // stall the LDS / scalar-memory counter the kernel's own waits count on.  seg[0] = wait + barrier, seg[1] = phase 0, seg[2] = phase 1.
  unsigned long long prof_seg[3] = {0ull, 0ull, 0ull}, prof_base = 0ull, prof_pend = 0ull;
  if constexpr ((ABL & 128) != 0) prof_base = prof_pend = __builtin_amdgcn_s_memtime();  // (the first two marks then add the prologue's time once: 1 / stages)
  auto prof_mark = [&](auto k_tag) __attribute__((always_inline)) {
    if constexpr ((ABL & 128) != 0) {
      constexpr int k = decltype(k_tag)::value;
      const unsigned long long t = prof_pend;
      prof_seg[(k + 1) % 3] += t - prof_base;
      prof_base = t;
      prof_pend = __builtin_amdgcn_s_memtime();
    }
  };
  constexpr int kYoung = (ST - 3) * PPW;
  constexpr bool p3 = P3;
  int c_slot = 0;  // ring slot of the stage being computed
  auto stage = [&](auto par_tag, bool first, bool relaxed, bool auxw) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value;
    const int n_slot = c_slot == ST - 1 ? 0 : c_slot + 1;
    prof_mark(std::integral_constant<int, 0>{});
    // stage s + 1 (a real one or a dummy of the stream's tail - nothing below needs to know) has landed once all but the ST - 3
    // younger stages (and, right after a tile boundary, the previous tile's stores, which are younger than every load waited
    // for here) are done
    // (auxw: the early auxiliary loads of this tile are among the younger loads that may stay outstanding, see aux_request)
    if (relaxed && auxw) wait_vm<kYoung + Cfg::STORES + AUXN>();
    else if (relaxed) wait_vm<kYoung + Cfg::STORES>();
    else if (auxw) wait_vm<kYoung + AUXN>();
    else wait_vm<kYoung>();
    if (!(dbg & 4)) __builtin_amdgcn_s_barrier();
    prof_mark(std::integral_constant<int, 1>{});
    // ---- phase 0: B lo of this stage + raw A of the next -> registers; hi.hi and lo.hi products; the ring refill in between
    read_b(bl, b_lo + (unsigned)(c_slot * STAGE));
    read_a((unsigned)(n_slot * STAGE));
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NM0 = 2 * TI * TJ;
    constexpr int STRIDE = NM0 / (PPW + 1) > 0 ? NM0 / (PPW + 1) : 1;
    // the hi.hi products of a tile's first stage start the accumulators from zero: two straight-line MFMA sequences, ONE branch
    auto phase0 = [&](auto first_tag) __attribute__((always_inline)) {
      constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) {
            const int idx = (h * TI + i) * TJ + j;
            if (!(dbg & 32)) {
              if (h == 0) {
                if (FIRST) acc[i][j] = mfma16<F16>(bh[PAR][j], ah[PAR][i], zero16);
                else acc[i][j] = mfma16<F16>(bh[PAR][j], ah[PAR][i], acc[i][j]);
              } else if (p3) {
                acc[i][j] = mfma16<F16>(bh[PAR][j], al[PAR][i], acc[i][j]);
              }
            } else if (FIRST && h == 0) {
              acc[i][j] = zero16;
            }
            if (idx % STRIDE == 0 && idx / STRIDE <= PPW) {
              __builtin_amdgcn_sched_barrier(0);
              issue_u(idx / STRIDE);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
    };
    if (PAR == 0 && first) phase0(std::true_type{});
    else phase0(std::false_type{});
    if constexpr (NM0 / STRIDE <= PPW) {  // (tiles with fewer MFMAs than pieces: the rest of the refill goes out here)
#pragma unroll
      for (int u = (NM0 - 1) / STRIDE + 1; u <= PPW; ++u) issue_u(u);
    }
    wait_ab(bl);
    prof_mark(std::integral_constant<int, 2>{});
    // ---- phase 1: B hi of the next stage -> the other register set; hi.lo products; the next stage's A split in between
    read_b(bh[PAR ^ 1], b_hi + (unsigned)(n_slot * STAGE));
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NM1 = TI * TJ, QS = 4 * TI;  // MFMAs of this phase, split quarters to place
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int idx = i * TJ + j;
        if (p3 && !(dbg & 32)) acc[i][j] = mfma16<F16>(bl[j], ah[PAR][i], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qq = idx * QS / NM1; qq < (idx + 1) * QS / NM1; ++qq) split_q(qq / 4, qq % 4);
        __builtin_amdgcn_sched_barrier(0);
      }
    split_commit(ah[PAR ^ 1], al[PAR ^ 1]);
    wait_b(bh[PAR ^ 1]);
    c_slot = n_slot;
  };

  // ---------------- priming: stage 0's operands into register set 0 ----------------
  wait_vm<(ST - 2) * PPW>();  // stage 0 landed (ST - 1 stages, real or dummy, were issued)
  __builtin_amdgcn_s_barrier();
  read_b(bh[0], b_hi);
  read_a(0u);
  wait_ab(bh[0]);
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) split_q(i, q);
  split_commit(ah[0], al[0]);

  const int my_tiles = (tiles - w + G - 1) / G;
  // (an epilogue that still loads an operand itself - add AND mask, scalar stores - waits for everything anyway)
  const bool relax_ok = p.vec_store && !(p.mask && p.add);
  // the auxiliary operand is requested at the top of stage s_aux = nst - (ST - 1), rounded down to an even stage (the stage loop
  // runs in register-set pairs), 0 for short reductions; (nst - s_aux) stages are issued between the request and the epilogue
  const int s_aux = aux_early ? (nst > ST - 1 ? (nst - (ST - 1)) & ~1 : 0) : nst + 2;
  typedef std::integral_constant<int, 0> P0;
  typedef std::integral_constant<int, 1> P1;
#pragma unroll 1
  for (int t = 0; t < my_tiles; ++t) {
    const int tile = w + t * G;
    const int bi = tile / tpb, rem = tile - bi * tpb;
    const EpiCtx ec{(rem / n_tiles) * BM, (rem % n_tiles) * BN, bi};
    // the previous tile's stores are younger than the loads of the stages that were in flight when they were issued: the first
    // ST - 2 stages of a tile may leave them outstanding (vector stores only: their number is the immediate's)
    const bool rx = t > 0 && relax_ok;
    // stages in pairs: even stages compute from register set 0 and prefetch into set 1, odd stages the other way round
#pragma unroll 1
    for (int s = 0; s < nst; s += 2) {
      if (s == s_aux) aux_request(ec);
      // aux window: while the request is younger than the stage being waited for (s - s_aux <= ST - 3)
      stage(P0{}, s == 0, rx && s < ST - 2, s >= s_aux && s - s_aux <= ST - 3);
      if (s + 1 < nst) stage(P1{}, false, rx && s + 1 < ST - 2, s + 1 >= s_aux && s + 1 - s_aux <= ST - 3);
    }
    if (nst & 1) {  // an odd number of stages: the next tile's first operands were prefetched into set 1 - hand them to set 0
#pragma unroll
      for (int i = 0; i < TI; ++i) { ah[0][i] = ah[1][i]; al[0][i] = al[1][i]; }
#pragma unroll
      for (int j = 0; j < TJ; ++j) bh[0][j] = bh[1][j];
    }
    if (aux_early) wait_vm_upto((nst - s_aux) * PPW);  // the auxiliary operand has landed; the stages issued since may stay in flight
    epilogue(ec);
  }
  wait_vm<0>();  // the dummy stages of the stream's tail are still landing: no LDS-DMA may outlive its workgroup's LDS allocation
  if constexpr ((ABL & 128) != 0) {
    if (p.prof && lane == 0) {
      unsigned long long* o = p.prof + ((long long)blockIdx.x * NW + wave) * 4;
      o[0] = prof_seg[0]; o[1] = prof_seg[1]; o[2] = prof_seg[2]; o[3] = (unsigned long long)my_tiles * nst;
    }
  }
  combo_ts_end(p.ts);
}

int g_force_tile = 0;  // combo_gemm_nt_x3_tile: 0 = planner, 1 wide, 2 mid, 3 skinny (tests, tools)

int n_cu_cached3() {
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return cus > 0 ? cus : 256;
  }();
  const int lim = combo_cu_limit();  // (abi.hip: a caller running two launch chains side by side hands each a share of the CUs)
  return lim > 0 && lim < n_cu ? lim : n_cu;
}

unsigned long long* g_prof3 = nullptr;  // combo_gemm_nt_x3_prof_buffer

int dbg_bits3() {
  static const int d = [] { const char* e = getenv("COMBO_NT3_DBG"); return e ? atoi(e) : 0; }();
  return d;
}

template <bool CONV, typename Cfg, bool P3, int ABL, bool AUX, bool F16 = false>
int launch_inst3x(const N3Args& a, int grid, hipStream_t stream) {
  static ComboDevFlag attr;
  if (!attr.is_set()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt3_kernel<CONV, Cfg, P3, ABL, AUX, F16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    if (e != hipSuccess) return (int)e;
    attr.mark();
  }
  hipLaunchKernelGGL((gemm_nt3_kernel<CONV, Cfg, P3, ABL, AUX, F16>), dim3((unsigned)grid), dim3(Cfg::NW * 64), Cfg::LDS, stream, a);
  return (int)hipGetLastError();
}

template <bool CONV, typename Cfg, bool P3, int ABL, bool F16 = false>
int launch_inst3(const N3Args& a, int grid, hipStream_t stream) {
  // the early-auxiliary-operand instance: launches with a vector-store epilogue that adds or masks (not the ablation builds, not
  // the 8-wave convolution tile: no registers for it)
  if constexpr (ABL == 0 && !(CONV && Cfg::NW == 8)) {
    if (a.vec_store && (a.add || a.mask)) return launch_inst3x<CONV, Cfg, P3, ABL, true, F16>(a, grid, stream);
  }
  return launch_inst3x<CONV, Cfg, P3, ABL, false, F16>(a, grid, stream);
}

template <bool CONV, typename Cfg>
int launch_cfg3(N3Args a, hipStream_t stream) {
  const long long tiles = ((a.M + Cfg::BM - 1LL) / Cfg::BM) * ((a.N + Cfg::BN - 1LL) / Cfg::BN) * a.batch;
  if (tiles > 0x7fffffffLL) return COMBO_EINVAL;
  const long long slots = n_cu_cached3();  // one persistent workgroup per CU
  const int grid = (int)(tiles < slots ? tiles : slots);
  a.ts = combo_timing_next_slot(a.products == 3 ? COMBO_TS_GEMM_X3 : COMBO_TS_GEMM_BF16, 2.0 * a.M * a.N * a.K * a.batch,
                                4.0 * a.batch * ((double)a.M * (CONV ? a.cg.Cin : a.K) + (double)a.N * a.K + (double)a.M * a.N * (1 + (a.mask ? 1 : 0) + (a.add ? 1 : 0))));
  if constexpr (!CONV && std::is_same<Cfg, NWide>::value) {  // the ablation instances (COMBO_NT3_DBG, tools/bench_nt3.py)
    if (a.products == 3) {
      switch (a.dbg) {
        case 1: return launch_inst3<CONV, Cfg, true, 1>(a, grid, stream);
        case 2: return launch_inst3<CONV, Cfg, true, 2>(a, grid, stream);
        case 4: return launch_inst3<CONV, Cfg, true, 4>(a, grid, stream);
        case 8: return launch_inst3<CONV, Cfg, true, 8>(a, grid, stream);
        case 16: return launch_inst3<CONV, Cfg, true, 16>(a, grid, stream);
        case 32: return launch_inst3<CONV, Cfg, true, 32>(a, grid, stream);
        case 41: return launch_inst3<CONV, Cfg, true, 41>(a, grid, stream);
        case 63: return launch_inst3<CONV, Cfg, true, 63>(a, grid, stream);
        case 128: return launch_inst3<CONV, Cfg, true, 128>(a, grid, stream);
        default: break;
      }
    }
  }
  if (a.products == 3 && a.f16) return launch_inst3<CONV, Cfg, true, 0, true>(a, grid, stream);  // fp16 pieces: 3 products only
  if (a.products == 3) return launch_inst3<CONV, Cfg, true, 0>(a, grid, stream);
  return launch_inst3<CONV, Cfg, false, 0>(a, grid, stream);
}

template <bool CONV>
int launch_one3(N3Args a, hipStream_t stream, int cfg) {
  if (cfg == 4) return launch_cfg3<CONV, NTall>(a, stream);
  if (cfg == 3) return launch_cfg3<CONV, NSkinny>(a, stream);
  if (cfg == 2) return launch_cfg3<CONV, NMid>(a, stream);
  return launch_cfg3<CONV, NWide>(a, stream);
}

// Tile choice: the load of the busiest CU (tiles are dealt round-robin to one workgroup per CU), with the measured per-MAC
// cost of the smaller tiles; when the last round of wide tiles is mostly empty the call is split by rows into whole rounds of
// large tiles + the remaining rows on small ones (as gemm_f32.hip).
template <bool CONV>
int launch_nt3(N3Args a, hipStream_t stream) {
  if (g_force_tile) return launch_one3<CONV>(a, stream, g_force_tile);
  const long long cus = n_cu_cached3();
  const int bm[5] = {0, 256, 128, 64, 256}, bn[5] = {0, 128, 128, 64, 64};
  auto tiles = [&](int c, long long rows) { return ((rows + bm[c] - 1) / bm[c]) * ((a.N + bn[c] - 1LL) / bn[c]) * a.batch; };
  const double eff[5] = {0.0, 1.0, 1.13, 1.8, 1.25};  // measured per-MAC cost of a round (tools/bench_nt3.py --shapes round)
  auto load = [&](int c, long long rows) { return (double)((tiles(c, rows) + cus - 1) / cus) * bm[c] * bn[c] * eff[c]; };
  int best = 1;
  double best_cost = load(1, a.M);
  for (int c = 2; c <= 4; ++c)
    if (load(c, a.M) < best_cost) { best = c; best_cost = load(c, a.M); }
  long long rows_main = a.M;
  int cfg_rest = 0;
  if (!CONV && a.batch == 1) {
    const double penalty = 1.0e6 / a.K;  // a second launch: ~3 us of a CU at this kernel's rate, in units of MACs / K
    for (int c = 1; c <= 2; ++c) {
      const long long tn = (a.N + bn[c] - 1LL) / bn[c], tm = (a.M + bm[c] - 1LL) / bm[c];
      const long long rounds = tm * tn / cus;
      if (rounds < 1) continue;
      const long long tm_main = rounds * cus / tn, rm = tm_main * bm[c];
      if (rm <= 0 || rm >= a.M) continue;
      for (int rr = c + 1; rr <= 3; ++rr) {
        const double cost = load(c, rm) + load(rr, a.M - rm) + penalty;
        if (cost < best_cost * 0.97) { best = c; best_cost = cost; rows_main = rm; cfg_rest = rr; }
      }
    }
  }
  if (rows_main >= a.M) return launch_one3<CONV>(a, stream, best);
  N3Args m = a, rs = a;
  m.M = (int)rows_main;
  m.c_bytes = (int)(((m.M - 1LL) * a.ldc + a.N) * 4);
  if (int e = launch_one3<CONV>(m, stream, best)) return e;
  rs.A = a.A + rows_main * a.lda;
  rs.C = a.C + rows_main * a.ldc;
  if (a.mask) rs.mask = a.mask + rows_main * a.ldc;
  if (a.add) rs.add = a.add + rows_main * a.ldc;
  rs.M = (int)(a.M - rows_main);
  rs.c_bytes = (int)(((rs.M - 1LL) * a.ldc + a.N) * 4);
  return launch_one3<CONV>(rs, stream, cfg_rest);
}

}  // namespace

/* Forces the tile shape of the combo_gemm_nt_x3_* / combo_conv3x3_nhwc_x3_* launches that FOLLOW (0 = the planner's choice, 1 =
 * 256 x 128, 2 = 128 x 128, 3 = 64 x 64, 4 = 256 x 64): every configuration must give the same result (tests), and tools sweep them.  Returns
 * the previous value.  Host-side state, read when a launch is issued. */
extern "C" int combo_gemm_nt_x3_tile(int cfg) {
  const int prev = g_force_tile;
  if (cfg >= 0 && cfg <= 4) g_force_tile = cfg;
  return prev;
}

/* COMBO_NT3_DBG=128 (the instrumented instance of the 8-wave tile): where the launches that follow write, per (workgroup, wave), the cycle sums
 * of a stage's three segments + the stage count (4 x u64 each; 256 workgroups x 8 waves); NULL = off.  tools/prof_nt3_stage.py */
extern "C" int combo_gemm_nt_x3_prof_buffer(unsigned long long* buf) {
  g_prof3 = buf;
  return 0;
}

int combo_nt3_launch(const float* A, long long lda, const float* Bimg, long long ldb, const float* bias, const float* mask, float* C,
                     long long ldc, long long M, int N, int K, int relu, int products, int batch, long long sA, long long sB,
                     long long sC, const combo_nt3_conv* conv, int force_cfg, combo_stream_t stream, const float* add) {
  if (!A || !Bimg || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || K % kBK != 0 || lda % 4 != 0 || ldb % 4 != 0 || sA % 4 != 0 ||
      sB % 4 != 0 || ((uintptr_t)A & 15) || ((uintptr_t)Bimg & 15) || M > 0x7fffffffLL || ((M - 1) * ldc + N) * 4 >= 0x7ffffff0LL ||
      (bias && N > kMaxBiasN) || (products != 1 && products != 3 && products != COMBO_PRODUCTS_F16X3 && products != COMBO_PRODUCTS_F16X3_UNSCALED))
    return COMBO_EINVAL;
  const int f16 = products == COMBO_PRODUCTS_F16X3 || products == COMBO_PRODUCTS_F16X3_UNSCALED ? 1 : 0;  // 3 products on fp16 pieces
  const float unscale = products == COMBO_PRODUCTS_F16X3 ? kF16Unscale : 1.0f;
  if (f16) products = 3;
  const int vec = (N % 4 == 0 && ldc % 4 == 0 && sC % 4 == 0 && !((uintptr_t)C & 15) && (!mask || !((uintptr_t)mask & 15)) && (!add || !((uintptr_t)add & 15))) ? 1 : 0;
  N3Args a{A, lda, Bimg, ldb, bias, mask, add, C, ldc, (int)M, N, K, relu, (int)(((M - 1) * ldc + N) * 4), batch, vec, dbg_bits3(), products,
           f16, unscale, sA, sB, sC, conv ? *conv : combo_nt3_conv{1, 1, K, 0, 1, 1}, nullptr, g_prof3};
  if (force_cfg) return conv ? launch_one3<true>(a, (hipStream_t)stream, force_cfg) : launch_one3<false>(a, (hipStream_t)stream, force_cfg);
  return conv ? launch_nt3<true>(a, (hipStream_t)stream) : launch_nt3<false>(a, (hipStream_t)stream);
}

namespace {
// out[m, n] = epilogue(sum_z part[z, m, n]): + bias[n], + add[m, n], ReLU, mask[m, n] > 0 ? . : 0 - the
// epilogue of the unsplit kernel; finishes a split-K GEMM / convolution; N % 4 == 0, fixed summation order
__global__ void __launch_bounds__(256)
nt3_splitk_finish_kernel(const float* __restrict__ part, int splits, long long M, int N, const float* __restrict__ mask,
                         float* __restrict__ out, long long ldc, const float* __restrict__ bias, int relu, const float* __restrict__ add) {
  const long long n4 = (long long)M * (N >> 2);
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= n4) return;
  const long long mrow = i / (N >> 2);
  const int c = (int)(i - mrow * (N >> 2)) * 4;
  float4 a = reinterpret_cast<const float4*>(part)[i];
  for (int z = 1; z < splits; ++z) {
    const float4 b = reinterpret_cast<const float4*>(part + (long long)z * M * N)[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  if (bias) {
    const float4 bv = *reinterpret_cast<const float4*>(bias + c);
    a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w;
  }
  if (add) {
    const float4 dv = *reinterpret_cast<const float4*>(add + mrow * ldc + c);
    a.x += dv.x; a.y += dv.y; a.z += dv.z; a.w += dv.w;
  }
  if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
  if (mask) {
    const float4 mk = *reinterpret_cast<const float4*>(mask + mrow * ldc + c);
    a.x = mk.x > 0.f ? a.x : 0.f; a.y = mk.y > 0.f ? a.y : 0.f; a.z = mk.z > 0.f ? a.z : 0.f; a.w = mk.w > 0.f ? a.w : 0.f;
  }
  *reinterpret_cast<float4*>(out + mrow * ldc + c) = a;
}
}  // namespace

/* Split-K plan of an input-gradient GEMM dX[M, N] = dY[M, K] . W: the number of K slices (1 = do not split).  A long reduction
 * with few output tiles leaves most CUs idle (the decoder FFN's linear1: 4000 x 2048 -> 256 is 64 tiles of 128 x 128 on 256 CUs,
 * 63 us where the whole operand set is 39 MB; the backbones' res5 / res4 1x1 layers likewise): the K slices run as the batch
 * entries of ONE launch into [splits, M, N] partials, a second small launch sums them in a fixed order (and applies the ReLU
 * mask).  Same arithmetic as the unsplit kernel up to re-association of the K sum. */
extern "C" int combo_gemm_nt_x3_splitk_plan(int M, int N, int K) {
  if (K < 512 || N % 4 != 0 || M <= 0) return 1;
  const long long cus = n_cu_cached3();
  const long long tiles = ((M + 127LL) / 128) * ((N + 127LL) / 128);  // mid tiles
  if (tiles * 2 > cus) return 1;
  int s = (int)(cus / tiles);
  if (s > 8) s = 8;
  while (s > 1 && (K % (s * 32) != 0 || K / s < 128)) --s;
  return s < 1 ? 1 : s;
}

int nt3_split_launch(const float* A, long long lda, const float* Bimg, const float* bias, const float* add, const float* mask,
                     float* C, long long ldc, long long M, int N, int K, int relu, int splits, float* workspace,
                     const combo_nt3_conv* conv, combo_stream_t stream, int products) {
  if (splits < 2 || !workspace || K % (splits * 32) != 0 || N % 4 != 0 || ((uintptr_t)workspace & 15) || ((uintptr_t)C & 15) ||
      (add && ((uintptr_t)add & 15)) || (mask && ((uintptr_t)mask & 15)) || (bias && ((uintptr_t)bias & 15)) || ldc % 4 != 0 ||
      M * N > 0x7fffffffLL / 4)
    return COMBO_EINVAL;
  const int Ks = K / splits;
  // slice z: A columns [z Ks, (z + 1) Ks) (element offset z Ks; a convolution: taps z * tap_step ...), image rows keep their pitch K
  // and start z Ks floats in
  if (int e = combo_nt3_launch(A, lda, Bimg, K, nullptr, nullptr, workspace, N, M, N, Ks, 0, products, splits, conv ? 0 : Ks, Ks, M * N,
                               conv, 0, stream, nullptr))
    return e;
  const long long n4 = M * (N >> 2);
  hipLaunchKernelGGL(nt3_splitk_finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace,
                     splits, M, N, mask, C, ldc, bias, relu, add);
  return (int)hipGetLastError();
}

extern "C" int combo_gemm_nt_x3_splitk_f32(const float* A, long long lda, const float* Bimg, const float* mask, float* C,
                                           long long ldc, int M, int N, int K, int splits, float* workspace, combo_stream_t stream) {
  return nt3_split_launch(A, lda, Bimg, nullptr, nullptr, mask, C, ldc, M, N, K, 0, splits, workspace, nullptr, stream);
}

/* C = epilogue(A[M, K] . image[N, K]^T) with the 3-product split: v = acc (+ bias[n]) (+ add[m, n]: the residual branch / the other
 * gradient arriving at the operand's producer), ReLU when relu, v = mask[m, n] > 0 ? v : 0 last (the ReLU gradient of the layer
 * that produced the operand); add and mask have C's pitch, either may be NULL.  splits > 1 (combo_gemm_nt_x3_splitk_plan): K
 * slices as batch entries into workspace [splits, M, N], the epilogue rides in the finishing sum.  The forward pass and the
 * input gradients of the ResNet backbones' 1x1 convolutions (FrozenBN folded into the weights; detectron2's BottleneckBlock, cited
 * at models/maskformer_model.py:138,145 of the reference). */
extern "C" int combo_gemm_nt_x3_epi2_f32(const float* A, long long lda, const float* Bimg, const float* bias, const float* add,
                                         const float* mask, float* C, long long ldc, int M, int N, int K, int relu, int splits,
                                         float* workspace, combo_stream_t stream) {
  if (splits > 1) return nt3_split_launch(A, lda, Bimg, bias, add, mask, C, ldc, M, N, K, relu, splits, workspace, nullptr, stream);
  return combo_nt3_launch(A, lda, Bimg, K, bias, mask, C, ldc, M, N, K, relu, 3, 1, 0, 0, 0, nullptr, 0, stream, add);
}

/* The same with ONE auxiliary tensor: aux_mode 0 none (aux NULL), 1 add, 2 mask. */
extern "C" int combo_gemm_nt_x3_epi_f32(const float* A, long long lda, const float* Bimg, const float* bias, const float* aux,
                                        int aux_mode, float* C, long long ldc, int M, int N, int K, int relu, int splits,
                                        float* workspace, combo_stream_t stream) {
  if (aux_mode < 0 || aux_mode > 2 || (aux_mode != 0) != (aux != nullptr)) return COMBO_EINVAL;
  return combo_gemm_nt_x3_epi2_f32(A, lda, Bimg, bias, aux_mode == 1 ? aux : nullptr, aux_mode == 2 ? aux : nullptr, C, ldc, M, N, K, relu,
                                   splits, workspace, stream);
}

/* Tap split of a 3x3 implicit-GEMM convolution over M tokens: 1 (do not split), 3 (one kernel row per slice) or 9 (one tap per
 * slice) - few output tiles (res4 / res5: 7 840 / 1 960 tokens at 40 frames) leave most CUs idle otherwise. */
extern "C" int combo_conv3x3_x3_splitk_plan(long long M, int Cout, int Cin) {
  if (M <= 0 || Cout % 4 != 0 || Cin % 32 != 0) return 1;
  const long long cus = n_cu_cached3();
  const long long tiles = ((M + 127) / 128) * ((Cout + 127LL) / 128);
  if (tiles * 2 > cus) return 1;
  // unit: one tap of 128 channels on a 128 x 128 tile (~3 us of a CU); the finishing pass costs a launch + splits reads of [M, Cout]
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= 9; s *= 3) {
    const double rounds = (double)((tiles * s + cus - 1) / cus);
    const double cost = rounds * (9 / s) * (Cin / 128.0) + (s > 1 ? 1.0 + 0.5 * s * ((double)M * Cout / 2.0e6) : 0.0);
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}

/* Y = epilogue(conv(X, W)) over NHWC tokens, 3-product split: ksize 3 (zero padding 1) or 1 (padding 0), stride 1 or 2 (output
 * map ceil(H / 2) x ceil(W / 2): the stride-2 3x3 and shortcut convolutions of a ResNet stage's first block); image =
 * combo_presplit_bf16x2_* of W as [Cout, (ky, kx, cin)]; epilogue and splits (3x3: 1, 3 or 9 = combo_conv3x3_x3_splitk_plan over
 * the OUTPUT tokens; 1x1: 1) as combo_gemm_nt_x3_epi_f32; aux / Y rows are output tokens. */
extern "C" int combo_conv_nhwc_x3_epi_f32(const float* X, long long ldx, const float* Wimg, const float* bias, const float* aux,
                                          int aux_mode, float* Y, long long ldy, int B, int H, int W, int Cin, int Cout, int ksize,
                                          int stride, int relu, int splits, float* workspace, combo_stream_t stream) {
  if (B <= 0 || H <= 0 || W <= 0 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return COMBO_EINVAL;
  const int ho = (H + stride - 1) / stride, wo = (W + stride - 1) / stride, taps = ksize * ksize;
  const long long M = (long long)B * ho * wo;
  if (!X || !Wimg || !Y || Cin <= 0 || Cout <= 0 || Cin % kBK != 0 || ldx % 4 != 0 || ((uintptr_t)X & 15) || ((uintptr_t)Wimg & 15) ||
      M > 0x7fffffffLL / 4 || (long long)B * H * W > 0x7fffffffLL / 4 || aux_mode < 0 || aux_mode > 2 ||
      (aux_mode != 0) != (aux != nullptr) || splits < 1 || taps % splits != 0)
    return COMBO_EINVAL;
  combo_nt3_conv cg{H, W, Cin, taps / splits, stride, ksize == 3 ? 1 : 0};
  if (splits > 1)
    return nt3_split_launch(X, ldx, Wimg, bias, aux_mode == 1 ? aux : nullptr, aux_mode == 2 ? aux : nullptr, Y, ldy, M, Cout, taps * Cin,
                            relu, splits, workspace, &cg, stream);
  return combo_nt3_launch(X, ldx, Wimg, (long long)taps * Cin, bias, aux_mode == 2 ? aux : nullptr, Y, ldy, M, Cout, taps * Cin, relu, 3, 1,
                          0, 0, 0, &cg, 0, stream, aux_mode == 1 ? aux : nullptr);
}

extern "C" int combo_conv3x3_nhwc_x3_epi_f32(const float* X, long long ldx, const float* Wimg, const float* bias, const float* aux,
                                             int aux_mode, float* Y, long long ldy, int B, int H, int W, int Cin, int Cout, int relu,
                                             int splits, float* workspace, combo_stream_t stream) {
  return combo_conv_nhwc_x3_epi_f32(X, ldx, Wimg, bias, aux, aux_mode, Y, ldy, B, H, W, Cin, Cout, 3, 1, relu, splits, workspace, stream);
}
