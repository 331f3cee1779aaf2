// gemm_nt_f32: C[M,N] = A[M,K] . B[N,K]^T (+ bias) (+ ReLU) in EXACT fp32 on the matrix cores
// (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, one rounding per product - bitwise an fmaf chain), and the implicit-GEMM
// 3x3 convolution on the same loop (CONV = true).
//
// Why this kernel exists (round 2): the 3-product bf16 split of gemm_nt3.hip carries ~2^-17 relative error per product
// against fp32's 2^-24.  Every FORWARD dense layer of the head ends, a few layers later, in `sigmoid(logit) < 0.5` (the
// attention masks of the decoder, transformer_decoder.py:502-507): a cell whose logit lies within the error band of 0
// flips, and the flip perturbs that query in all following layers - 0.07 / 0.42 / 0.66 % of the mask logits of prediction
// heads 7 / 8 / 9 ended beyond the north-star's 1e-3 tolerance.  With fp32 GEMMs the same head has no outlier
// (tests/test_head_gpu.py).  So the forward path computes in fp32; the gradient GEMMs (no threshold downstream, tolerance
// 2e-3) keep the 3-product split, which is ~1.9x faster.
//
// Roofline: MFMA-bound.  The f32 MFMA issues once per 64 cycles per SIMD = 64 flop/clk/SIMD = 157.3 TFLOP/s on the chip
// (the fp32 vector rate, 1/16 of bf16); operand traffic is a rounding error next to that (12 ds_read_b128 and 6 LDS-DMA
// pieces per 64 MFMAs), so the only way to lose time is an idle matrix pipe.  v1 of this file (gemm_nt2's skeleton: two
// workgroups per CU taking turns) measured 48-74 % of the peak: equal tiles keep the two workgroups in phase, so both sit in
// their epilogue (128 scattered dword stores per lane) or at the per-stage barrier at the same time.  v2 makes ONE wave per
// SIMD keep its pipe busy on its own:
//   * one persistent workgroup per CU (4 waves, 512 registers each), LDS-DMA ring of BK = 16 stages as before (chunk swizzle
//     on the SOURCE address, counted vmcnt across a raw s_barrier);
//   * rolling half-stage operand prefetch: while the MFMAs of k = 0..7 of a stage issue, the k = 8..15 fragments are read
//     from LDS into the registers the previous half just freed (and vice versa, across stage and tile boundaries), so no
//     MFMA ever waits on an LDS read and the one barrier per stage falls between two MFMA blocks;
//   * the last half-stage of a tile runs sub-tile-major (4 dependent MFMAs per 32 x 32 sub-tile: the dependent latency of
//     this MFMA equals its issue interval) and stores the previous, finished sub-tile meanwhile: the epilogue hides behind
//     MFMAs, and the next tile's first fragments are already in registers when it ends;
//   * MFMA operands swapped (D = W . X^T): a lane then owns 4 consecutive n of one token, i.e. dwordx4 stores - 4 instead
//     of 16 store instructions per 32 x 32 sub-tile.
// Tile shapes (chosen per call by the smallest "busiest CU" load, launch_f32 below):
//   wide   256 x 128 (4 x 1 waves, 2 x 4 sub-tiles)   mid 128 x 128 (2 x 2 waves, 2 x 2)   skinny 64 x 64 (2 x 2 waves, 1 x 1)
//
// k order inside a stage: lane (row r, half g) holds k = 8g .. 8g+7 of its rows (two 16-byte chunks c = 0, 1); MFMA step
// t = 4c + e pairs k = t (g = 0) with k = 8 + t (g = 1) for both operands - any pairing is a valid contraction order.
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
constexpr int kMaxBiasN = 4096;  // the bias vector lives in LDS for the epilogue (ordinary global loads next to the LDS-DMA
                                 // stream make hipcc drain the whole ring with s_waitcnt vmcnt(0))

__device__ __attribute__((aligned(64))) float g_zero_row_f32[16];  // zero-initialised: the source of padded conv taps

struct ConvGeomF {
  int H, W, Cin;
};

template <int WM_, int WN_, int TI_, int TJ_, int ST_>
struct F32Cfg {
  static constexpr int WM = WM_, WN = WN_, TI = TI_, TJ = TJ_, ST = ST_;
  static constexpr int BM = WM * TI * 32, BN = WN * TJ * 32;
  static constexpr int A_BYTES = BM * kBK * 4, B_BYTES = BN * kBK * 4, STAGE = A_BYTES + B_BYTES;
  static constexpr int A_PIECES = A_BYTES / 1024, PIECES = STAGE / 1024, PPW = PIECES / 4, APW = A_PIECES / 4;
  static constexpr int RING = ST * STAGE;
  static constexpr int LDS = RING + kMaxBiasN * 4;  // ring + the bias vector (read by the epilogue with ds_read)
  static_assert(WM * WN == 4 && PIECES % 4 == 0 && A_PIECES % 4 == 0, "4 waves share the DMA pieces evenly");
  static_assert(ST >= 3 && (ST - 2) * PPW <= 63, "vmcnt is a 6-bit counter");
  static_assert(LDS <= 160 * 1024, "LDS budget of a CU");
};
typedef F32Cfg<4, 1, 2, 4, 5> FWide;     // 24 KiB stages, 120 KiB ring
typedef F32Cfg<2, 2, 2, 2, 8> FMid;      // 16 KiB stages, 128 KiB ring
typedef F32Cfg<2, 2, 1, 1, 16> FSkinny;  //  8 KiB stages, 128 KiB ring = a whole K = 256 panel in flight

struct F32Args {
  const float* A; long long lda;
  const float* B; long long ldb;
  const float* bias;      // [N] or nullptr
  float* C; long long ldc;
  int M, N, K, relu, c_bytes, batch, vec_store, dbg;  // dbg: ablation bits (COMBO_F32_DBG, tools/bench_f32.py): 1 no DMA, 2 no LDS reads, 4 no barrier, 8 no stores
  long long sA, sB, sC;   // batch strides (elements) of the INNER batch index bi % bdiv ...
  int bdiv;               // ... and of the outer index bi / bdiv (0: one-level batch, the outer strides are unused)
  long long sA2, sB2, sC2;
  ConvGeomF cg;
  unsigned long long* ts; // device-side timing slot (combo_common.h) or nullptr
};

template <bool CONV, typename Cfg>
__global__ void __launch_bounds__(256, 1)
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
  if (w >= tiles) {  // (grid <= tiles by construction; kept for safety: no workgroup may skip the barriers below)
    combo_ts_end(p.ts);
    return;
  }

  // the bias vector goes to LDS once (zero beyond N), before any LDS-DMA is in flight
  const unsigned bias_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)Cfg::RING;
  const int n_pad = min(kMaxBiasN, (N + 3) & ~3);  // only the N values that exist are staged (one pass for N <= 256)
  if (p.bias) {
    float* bl = reinterpret_cast<float*>(smem + Cfg::RING);
    for (int n = threadIdx.x; n < n_pad; n += 256) bl[n] = n < N ? p.bias[n] : 0.f;
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
  auto open_tile = [&]() __attribute__((always_inline)) {
    const int bi = i_tile / tpb, rem = i_tile - bi * tpb;
    const int b_in = p.bdiv ? bi % p.bdiv : bi, b_out = p.bdiv ? bi / p.bdiv : 0;
    const float* i_A = A + b_in * p.sA + b_out * p.sA2;
    const float* i_B = Bm + b_in * p.sB + b_out * p.sB2;
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
  // one DMA piece (1 KiB = 16 rows x 64 B, wave-uniform destination) of the stage under the issue cursor
  auto issue_piece = [&](auto u_tag) __attribute__((always_inline)) {
    constexpr int u = decltype(u_tag)::value;
    if (i_tile >= tiles || (p.dbg & 1)) return;
    char* st = smem + i_slot * STAGE;
    const int q = wave + 4 * u;  // q < A_PIECES: A rows, else B rows
    if constexpr (u < APW) {
      const long long a_off = CONV ? (long long)((i_tap / 3 - 1) * cg.W + (i_tap % 3 - 1)) * lda + i_cin0 : (long long)(i_s * kBK);
      const float* src = pa[u] + a_off;
      if (CONV) {
        const int c = p_chunk ^ (((q * 16 + p_row) >> 2) & 3);
        if (!((tap_ok[u] >> i_tap) & 1u)) src = g_zero_row_f32 + c * 4;
      }
      glds16(src, st + q * 1024);
    } else {
      glds16(pb[u - APW] + i_s * kBK, st + A_BYTES + (q - Cfg::A_PIECES) * 1024);
    }
  };
  auto issue_finish = [&]() __attribute__((always_inline)) {  // advance the cursor past the stage just issued
    if (i_tile >= tiles) return;
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
  auto issue_next = [&]() __attribute__((always_inline)) {
    issue_piece(std::integral_constant<int, 0>{});
    if constexpr (PPW > 1) issue_piece(std::integral_constant<int, 1>{});
    if constexpr (PPW > 2) issue_piece(std::integral_constant<int, 2>{});
    if constexpr (PPW > 3) issue_piece(std::integral_constant<int, 3>{});
    if constexpr (PPW > 4) issue_piece(std::integral_constant<int, 4>{});
    if constexpr (PPW > 5) issue_piece(std::integral_constant<int, 5>{});
    static_assert(PPW <= 6, "add the pieces of a larger stage here");
    issue_finish();
  };
  open_tile();
#pragma unroll 1
  for (int q = 0; q < ST - 1; ++q) issue_next();

  // ---------------- LDS read addresses: lane (row r, k-half g) reads chunks 2g, 2g+1 of its rows ----------------
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int r = lane & 31, g = lane >> 5;
  const int sw = (r >> 2) & 3;  // a wave's rows are (multiple of 32) + r: they all share the swizzle of r
  const unsigned a_c0 = lds0 + (unsigned)((wm * TI * 32 + r) * 64 + ((2 * g) ^ sw) * 16);
  const unsigned a_c1 = lds0 + (unsigned)((wm * TI * 32 + r) * 64 + ((2 * g + 1) ^ sw) * 16);
  const unsigned b_c0 = lds0 + (unsigned)(A_BYTES + (wn * TJ * 32 + r) * 64 + ((2 * g) ^ sw) * 16);
  const unsigned b_c1 = lds0 + (unsigned)(A_BYTES + (wn * TJ * 32 + r) * 64 + ((2 * g + 1) ^ sw) * 16);

  // operand fragments: fa[i][c] = token rows (MFMA src B), fb[j][c] = weight rows (MFMA src A); c = chunk = half of a stage
  f4v fa[TI][2], fb[TJ][2];
#pragma unroll
  for (int i = 0; i < TI; ++i) fa[i][0] = fa[i][1] = f4v{1.f, 1.f, 1.f, 1.f};  // (only observable under the no-LDS-read ablation)
#pragma unroll
  for (int j = 0; j < TJ; ++j) fb[j][0] = fb[j][1] = f4v{1.f, 1.f, 1.f, 1.f};
  auto read_half = [&](auto c_tag, unsigned so) __attribute__((always_inline)) {  // issue the ds_reads of chunk c of the stage in ring slot offset `so`
    constexpr int c = decltype(c_tag)::value;
    if (p.dbg & 2) return;
    const unsigned ab = (c == 0 ? a_c0 : a_c1) + so, bb = (c == 0 ? b_c0 : b_c1) + so;
#pragma unroll
    for (int i = 0; i < TI; ++i) fa[i][c] = i == 0 ? lds_read128<0>(ab) : lds_read128<2048>(ab);
#pragma unroll
    for (int j = 0; j < TJ; ++j)
      fb[j][c] = j == 0 ? lds_read128<0>(bb) : j == 1 ? lds_read128<2048>(bb) : j == 2 ? lds_read128<4096>(bb) : lds_read128<6144>(bb);
    __builtin_amdgcn_sched_barrier(0);  // the reads go out BEFORE the MFMAs that cover their latency
  };
  auto wait_half = [&](auto c_tag) __attribute__((always_inline)) {  // lgkmcnt(0) tied to the registers of chunk c (hipcc may not move their uses above it)
    constexpr int c = decltype(c_tag)::value;
    __builtin_amdgcn_sched_barrier(0);  // (and the MFMAs issued before the wait stay before it: they cover the read latency)
    if constexpr (TI == 2 && TJ == 4) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fa[0][c]), "+v"(fa[1][c]), "+v"(fb[0][c]), "+v"(fb[1][c]), "+v"(fb[2][c]), "+v"(fb[3][c]) : : "memory");
    } else if constexpr (TI == 2 && TJ == 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][c]), "+v"(fa[1][c]), "+v"(fb[0][c]), "+v"(fb[1][c]) : : "memory");
    } else {
      static_assert((TI == 2 && TJ == 4) || (TI == 2 && TJ == 2) || (TI == 1 && TJ == 1), "add the register list of a new wave tile here");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][c]), "+v"(fb[0][c]) : : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);  // (register-only MFMAs may not be hoisted above the inline-asm wait)
  };

  f32x16 acc[TI][TJ];

  // epilogue of one 32 x 32 sub-tile: lane (token r, half g) owns n = 8q + 4g + (0..3), q = 0..3 (operands swapped, see top)
  struct EpiCtx { int m_blk, n_blk; };
  auto store_sub = [&](auto ij_tag, const EpiCtx& ec, const __amdgpu_buffer_rsrc_t c_rsrc) __attribute__((always_inline)) {
    constexpr int IJ = decltype(ij_tag)::value, i = IJ / TJ, j = IJ % TJ;
    const int row = ec.m_blk + (wm * TI + i) * 32 + r;
    const int nb = ec.n_blk + (wn * TJ + j) * 32 + 4 * g;
    if (p.dbg & 8) {  // keep the accumulators alive without storing them
#pragma unroll
      for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(acc[i][j][e]));
      return;
    }
    if (p.vec_store) {
      f4v bv[4];
      if (p.bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = lds_read128<0>(bias_lds + (unsigned)min(nb + 8 * q, n_pad - 4) * 4u);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]) : : "memory");
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = nb + 8 * q;
        f4v v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        if (p.bias) v += bv[q];
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        // rows >= M fall outside the descriptor's range and are dropped; columns >= N are steered there as well
        unsigned off = n0 < N ? ((unsigned)row * (unsigned)p.ldc + (unsigned)n0) * 4u : 0xfffffff0u;
        if (p.dbg & 16) {  // ablation: the same bytes in a full-line pattern (8 rows x 128 B per instruction; WRONG layout)
          const int row2 = ec.m_blk + (wm * TI + i) * 32 + (lane >> 3) + 8 * q, n2 = ec.n_blk + (wn * TJ + j) * 32 + (lane & 7) * 4;
          off = n2 < N ? ((unsigned)row2 * (unsigned)p.ldc + (unsigned)n2) * 4u : 0xfffffff0u;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), c_rsrc, off, 0, 0);
      }
    } else {  // N or ldc not a multiple of 4 (the 3-wide class head): scalar stores
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = nb + 8 * (e >> 2) + (e & 3);
        float v = acc[i][j][e];
        if (p.bias) {
          float b;
          asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(bias_lds + (unsigned)min(n, n_pad - 1) * 4u) : "memory");
          v += b;
        }
        if (p.relu) v = fmaxf(v, 0.f);
        const unsigned off = n < N ? ((unsigned)row * (unsigned)p.ldc + (unsigned)n) * 4u : 0xfffffff0u;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), c_rsrc, off, 0, 0);
      }
    }
  };

  // one half-stage of MFMAs (k-steps 4c .. 4c+3).  FIRST: the first k-step of a tile starts from zero.  ISSUE: the LDS-DMA
  // pieces of the stage entering the ring are issued between the MFMAs (an MFMA occupies the pipe for 64 cycles = ~15 issue
  // slots; a DMA piece costs ~60 cycles of issue), so the matrix pipe never waits for the address arithmetic
  auto mfma_half = [&](auto c_tag, auto first_tag, auto issue_tag) __attribute__((always_inline)) {
    constexpr int c = decltype(c_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value, ISSUE = decltype(issue_tag)::value;
    constexpr int NM = 4 * TI * TJ;                       // MFMAs in this half
    constexpr int STRIDE = NM / (PPW + 1) > 0 ? NM / (PPW + 1) : 1;  // one piece every STRIDE MFMAs, then the cursor update
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          if (FIRST && c == 0 && e == 0) {
            f32x16 z;
#pragma unroll
            for (int q = 0; q < 16; ++q) z[q] = 0.f;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][c][e], fa[i][c][e], z, 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][c][e], fa[i][c][e], acc[i][j], 0, 0, 0);
          }
          if constexpr (ISSUE) {
            const int idx = (e * TI + i) * TJ + j;  // (compile-time after unrolling)
            if (idx % STRIDE == 0 && idx / STRIDE <= PPW) {
              __builtin_amdgcn_sched_barrier(0);
              const int u = idx / STRIDE;
              if (u == 0) issue_piece(std::integral_constant<int, 0>{});
              if (u == 1 && PPW > 1) issue_piece(std::integral_constant<int, (PPW > 1 ? 1 : 0)>{});
              if (u == 2 && PPW > 2) issue_piece(std::integral_constant<int, (PPW > 2 ? 2 : 0)>{});
              if (u == 3 && PPW > 3) issue_piece(std::integral_constant<int, (PPW > 3 ? 3 : 0)>{});
              if (u == 4 && PPW > 4) issue_piece(std::integral_constant<int, (PPW > 4 ? 4 : 0)>{});
              if (u == 5 && PPW > 5) issue_piece(std::integral_constant<int, (PPW > 5 ? 5 : 0)>{});
              if (u == PPW) issue_finish();
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
  };
  // the LAST half-stage of a tile, sub-tile-major: the 4 k-steps of a sub-tile issue back to back (the dependent-accumulator
  // latency of this MFMA equals its issue interval), and the finished PREVIOUS sub-tile is stored meanwhile - the epilogue
  // hides behind the MFMAs except for the last sub-tile's four stores
  auto mfma_tail = [&](const EpiCtx& ec, const __amdgpu_buffer_rsrc_t c_rsrc) __attribute__((always_inline)) {
    auto sub = [&](auto ij_tag) __attribute__((always_inline)) {
      constexpr int IJ = decltype(ij_tag)::value, i = IJ / TJ, j = IJ % TJ;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][1][e], fa[i][1][e], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (IJ > 0) store_sub(std::integral_constant<int, IJ - 1>{}, ec, c_rsrc);
      __builtin_amdgcn_sched_barrier(0);
    };
    sub(std::integral_constant<int, 0>{});
    if constexpr (TI * TJ >= 4) {
      sub(std::integral_constant<int, 1>{}); sub(std::integral_constant<int, 2>{}); sub(std::integral_constant<int, 3>{});
    }
    if constexpr (TI * TJ == 8) {
      sub(std::integral_constant<int, 4>{}); sub(std::integral_constant<int, 5>{}); sub(std::integral_constant<int, 6>{});
      sub(std::integral_constant<int, 7>{});
    }
    store_sub(std::integral_constant<int, TI * TJ - 1>{}, ec, c_rsrc);
  };

  // ---------------- the stage stream ----------------
  // State at the top of stage gs: its chunk 0 is in fa/fb[.][0]; stages up to gs + ST - 2 are landed / in flight.
  //   half 0: read chunk 1 of gs; MFMAs k = 0..7 with the DMA pieces of stage gs + ST - 1 in between (its ring slot held
  //           stage gs - 1, which every wave finished reading before the previous barrier)
  //   middle: stage gs + 1 must have landed for everybody: counted vmcnt (ST - 2 younger stages stay in flight) + barrier
  //   half 1: read chunk 0 of gs + 1; MFMAs k = 8..15
  constexpr int kYoung = (ST - 2) * PPW;
  static_assert(kYoung <= 63, "vmcnt is a 6-bit counter");
  if (issued == ST - 1) wait_vm<kYoung>();  // priming: stage 0 landed, ST - 2 stages stay in flight
  else wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  int c_slot = 0;  // ring slot of the stage being computed
  int left = ((tiles - w + G - 1) / G) * nst;  // stages this workgroup still has to compute
  read_half(std::integral_constant<int, 0>{}, 0u);
  wait_half(std::integral_constant<int, 0>{});

  auto mid_stage = [&]() __attribute__((always_inline)) {
    --left;
    if (left > 0) {
      // in steady state exactly ST - 2 younger stages are in flight; when the cursor has run out, drain everything
      // (epilogue stores in between only make the wait longer, never shorter)
      if (i_tile < tiles || left >= ST - 1) wait_vm<kYoung>();
      else wait_vm<0>();
      if (!(p.dbg & 4)) __builtin_amdgcn_s_barrier();
      c_slot = c_slot == ST - 1 ? 0 : c_slot + 1;
      read_half(std::integral_constant<int, 0>{}, (unsigned)(c_slot * STAGE));
    }
  };

  const int my_tiles = (tiles - w + G - 1) / G;
#pragma unroll 1
  for (int t = 0; t < my_tiles; ++t) {
    const int tile = w + t * G;
    const int bi = tile / tpb, rem = tile - bi * tpb;
    const EpiCtx ec{(rem / n_tiles) * BM, (rem % n_tiles) * BN};
    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.C + (p.bdiv ? (bi % p.bdiv) * p.sC + (bi / p.bdiv) * p.sC2 : bi * p.sC), 0, p.c_bytes, 0x00020000);
    // ---- stage 0 (peeled: its first k-step starts the accumulators from zero)
    read_half(std::integral_constant<int, 1>{}, (unsigned)(c_slot * STAGE));
    mfma_half(std::integral_constant<int, 0>{}, std::true_type{}, std::true_type{});
    wait_half(std::integral_constant<int, 1>{});
    mid_stage();
    if (nst == 1) mfma_tail(ec, c_rsrc);
    else mfma_half(std::integral_constant<int, 1>{}, std::false_type{}, std::false_type{});
    if (left > 0) wait_half(std::integral_constant<int, 0>{});
    // ---- stages 1 .. nst-2
#pragma unroll 1
    for (int s = 1; s < nst - 1; ++s) {
      read_half(std::integral_constant<int, 1>{}, (unsigned)(c_slot * STAGE));
      mfma_half(std::integral_constant<int, 0>{}, std::false_type{}, std::true_type{});
      wait_half(std::integral_constant<int, 1>{});
      mid_stage();
      mfma_half(std::integral_constant<int, 1>{}, std::false_type{}, std::false_type{});
      wait_half(std::integral_constant<int, 0>{});
    }
    // ---- last stage (its second half carries the epilogue)
    if (nst > 1) {
      read_half(std::integral_constant<int, 1>{}, (unsigned)(c_slot * STAGE));
      mfma_half(std::integral_constant<int, 0>{}, std::false_type{}, std::true_type{});
      wait_half(std::integral_constant<int, 1>{});
      mid_stage();
      mfma_tail(ec, c_rsrc);
      if (left > 0) wait_half(std::integral_constant<int, 0>{});
    }
  }
  combo_ts_end(p.ts);
}

int n_cu_cached() {
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return cus > 0 ? cus : 256;
  }();
  const int lim = combo_cu_limit();  // (abi.hip: a caller running two launch chains side by side hands each a share of the CUs)
  return lim > 0 && lim < n_cu ? lim : n_cu;
}

template <bool CONV, typename Cfg>
int launch_cfg(F32Args a, hipStream_t stream) {
  static ComboDevFlag attr;
  if (!attr.is_set()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_f32_kernel<CONV, Cfg>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    if (e != hipSuccess) return (int)e;
    attr.mark();
  }
  const long long tiles = ((a.M + Cfg::BM - 1LL) / Cfg::BM) * ((a.N + Cfg::BN - 1LL) / Cfg::BN) * a.batch;
  if (tiles > 0x7fffffffLL) return COMBO_EINVAL;
  const long long slots = n_cu_cached();  // one persistent workgroup per CU
  const int grid = (int)(tiles < slots ? tiles : slots);
  a.ts = combo_timing_next_slot(COMBO_TS_GEMM_F32, 2.0 * a.M * a.N * a.K * a.batch,
                                4.0 * a.batch * ((double)a.M * (CONV ? a.K / 9 : a.K) + (double)a.N * a.K + (double)a.M * a.N));
  hipLaunchKernelGGL((gemm_nt_f32_kernel<CONV, Cfg>), dim3((unsigned)grid), dim3(256), Cfg::LDS, stream, a);
  return (int)hipGetLastError();
}

// Tile choice.  The kernel is MFMA-bound, so its time is the load of the busiest CU: tiles are dealt round-robin to one
// workgroup per CU, each costing BM*BN*K MACs.  Measured on MI355X (tools/bench_f32.py with COMBO_F32_TILE = 1 / 2 / 3 over the
// head's shapes, profiles/r02_gemm_f32_tiles.txt): the smallest "ceil(tiles / CUs) * BM * BN * (measured time per MAC of the
// tile shape)" wins - small tiles pay more per MAC (tile boundaries; 64 x 64 tiles pull 16 flop per operand byte from L2) and
// only win where they fill the chip better (the decoder's 4000-token layers: 252 tiles instead of 32).
// Wave quantisation: 1288 wide tiles on 256 CUs are 5.03 rounds = 6 rounds of time.  When the last round is mostly empty the
// call is SPLIT BY ROWS into two launches: whole rounds of large tiles, then the remaining rows on smaller tiles (41160 x 1024:
// 1280 wide tiles + 200 rows as 64 skinny tiles = 5.1 rounds of work instead of 6).
struct Plan {
  int cfg;             // 1 wide, 2 mid, 3 skinny
  long long rows_main; // rows of the first launch (== M: single launch)
  int cfg_rest;
  double cost;         // busiest-CU MACs / K
};

template <bool CONV>
int launch_one(F32Args a, hipStream_t stream, int cfg) {
  if (cfg == 3) return launch_cfg<CONV, FSkinny>(a, stream);
  if (cfg == 2) return launch_cfg<CONV, FMid>(a, stream);
  return launch_cfg<CONV, FWide>(a, stream);
}

template <bool CONV>
int launch_f32(F32Args a, hipStream_t stream) {
  static const int force = [] { const char* e = getenv("COMBO_F32_TILE"); return e ? atoi(e) : 0; }();  // 1 wide, 2 mid, 3 skinny (A/B)
  const int no_split = 0;
  if (force) return launch_one<CONV>(a, stream, force);
  const long long cus = n_cu_cached();
  const int bm[4] = {0, 256, 128, 64}, bn[4] = {0, 128, 128, 64};
  auto tiles = [&](int c, long long rows) { return ((rows + bm[c] - 1) / bm[c]) * ((a.N + bn[c] - 1LL) / bn[c]) * a.batch; };
  // measured time per MAC relative to the wide tile (per-tile overheads weigh more on small tiles): 232 us / 6 rounds wide,
  // 231 us / 11 rounds mid, 272 us / 41 rounds skinny at 41160 x 256 -> 1024
  const double eff[4] = {0.0, 1.0, 1.085, 1.37};
  auto load = [&](int c, long long rows) { return (double)((tiles(c, rows) + cus - 1) / cus) * bm[c] * bn[c] * eff[c]; };
  Plan best{1, a.M, 0, load(1, a.M)};
  for (int c = 2; c <= 3; ++c)
    if (load(c, a.M) < best.cost) best = Plan{c, a.M, 0, load(c, a.M)};
  if (!CONV && a.batch == 1 && !no_split) {
    // a second launch costs ~6 us: ~1.2 M MACs of a CU at this kernel's rate, in units of MACs / K
    const double penalty = 1.2e6 / a.K;
    for (int c = 1; c <= 2; ++c) {
      const long long tn = (a.N + bn[c] - 1LL) / bn[c], tm = (a.M + bm[c] - 1LL) / bm[c];
      const long long rounds = tm * tn / cus;
      if (rounds < 1) continue;
      const long long tm_main = rounds * cus / tn, rows_main = tm_main * bm[c];
      if (rows_main <= 0 || rows_main >= a.M) continue;
      for (int r = c + 1; r <= 3; ++r) {
        const double cost = load(c, rows_main) + load(r, a.M - rows_main) + penalty;
        if (cost < best.cost * 0.97) best = Plan{c, rows_main, r, cost};
      }
    }
  }
  if (best.rows_main >= a.M) return launch_one<CONV>(a, stream, best.cfg);
  F32Args m = a, r = a;
  m.M = (int)best.rows_main;
  m.c_bytes = (int)(((m.M - 1LL) * a.ldc + a.N) * 4);
  if (int e = launch_one<CONV>(m, stream, best.cfg)) return e;
  r.A = a.A + best.rows_main * a.lda;
  r.C = a.C + best.rows_main * a.ldc;
  r.M = (int)(a.M - best.rows_main);
  r.c_bytes = (int)(((r.M - 1LL) * a.ldc + a.N) * 4);
  return launch_one<CONV>(r, stream, best.cfg_rest);
}

bool args_ok(const float* A, long long lda, const float* B, long long ldb, const float* C, long long ldc, long long M, int N, int K,
             int batch) {
  return A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && K % kBK == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
         !((uintptr_t)A & 15) && !((uintptr_t)B & 15) && M <= 0x7fffffffLL && ((M - 1) * ldc + N) * 4 < 0x7ffffff0LL;
}

int dbg_bits() {
  static const int d = [] { const char* e = getenv("COMBO_F32_DBG"); return e ? atoi(e) : 0; }();
  return d;
}

int vec_ok(const float* C, long long ldc, long long sC, int N, const float* bias) {
  (void)bias;
  return (N % 4 == 0 && ldc % 4 == 0 && sC % 4 == 0 && !((uintptr_t)C & 15)) ? 1 : 0;
}

}  // namespace

extern "C" int combo_gemm_nt_f32(const float* A, long long lda, const float* B, long long ldb, const float* bias, float* C,
                                 long long ldc, int M, int N, int K, int relu, combo_stream_t stream) {
  if (!args_ok(A, lda, B, ldb, C, ldc, M, N, K, 1) || (bias && N > kMaxBiasN)) return COMBO_EINVAL;
  F32Args a{A, lda, B, ldb, bias, C, ldc, M, N, K, relu, (int)(((M - 1LL) * ldc + N) * 4), 1, vec_ok(C, ldc, 0, N, bias), dbg_bits(), 0, 0, 0,
            0, 0, 0, 0, ConvGeomF{1, 1, K}, nullptr};
  return launch_f32<false>(a, (hipStream_t)stream);
}

namespace {
// out[m, n] = sum_z part[z, m, n] (+ bias[n]) (+ ReLU): finishes a split-K forward GEMM; N % 4 == 0, fixed summation order
__global__ void __launch_bounds__(256)
splitk_finish_kernel(const float* __restrict__ part, int splits, long long M, int N, const float* __restrict__ bias, int relu,
                     float* __restrict__ out, long long ldc) {
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
  if (bias) { a.x += bias[c]; a.y += bias[c + 1]; a.z += bias[c + 2]; a.w += bias[c + 3]; }
  if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
  *reinterpret_cast<float4*>(out + mrow * ldc + c) = a;
}
}  // namespace

/* Split-K plan of a forward GEMM: the number of K slices (1 = do not split).  A long reduction with few output tiles leaves
 * most CUs idle (decoder FFN linear2: 4000 x 2048 -> 256 is 64 tiles on 256 CUs; measured 54 us against 39 us for the
 * library); the K slices run as the batch entries of ONE launch into [splits, M, N] partials, a second tiny launch sums them
 * in a fixed order and applies bias / ReLU.  Summation order differs from the unsplit kernel by re-association only. */
extern "C" int combo_gemm_nt_splitk_plan(int M, int N, int K) {
  if (K < 1024 || N % 4 != 0) return 1;
  const long long cus = n_cu_cached();
  const long long tiles = ((M + 127LL) / 128) * ((N + 127LL) / 128);  // mid tiles
  if (tiles * 2 > cus) return 1;
  int s = (int)(cus / tiles);
  while (s > 1 && (K % (s * kBK) != 0 || K / s < 256)) --s;
  return s > 8 ? 8 : (s < 1 ? 1 : s);
}

extern "C" int combo_gemm_nt_splitk_f32(const float* A, long long lda, const float* B, long long ldb, const float* bias, float* C,
                                        long long ldc, int M, int N, int K, int relu, int splits, float* workspace,
                                        combo_stream_t stream) {
  if (splits < 2 || !workspace || K % (splits * kBK) != 0 || N % 4 != 0 || ((uintptr_t)workspace & 15) || ((uintptr_t)C & 15) ||
      ldc % 4 != 0 || !args_ok(A, lda, B, ldb, workspace, N, M, N, K / splits, splits) || (long long)M * N > 0x7fffffffLL / 4)
    return COMBO_EINVAL;
  const int Ks = K / splits;
  F32Args a{A, lda, B, ldb, nullptr, workspace, N, M, N, Ks, 0, (int)(((M - 1LL) * N + N) * 4), splits,
            vec_ok(workspace, N, (long long)M * N, N, nullptr), dbg_bits(), Ks, Ks, (long long)M * N, 0, 0, 0, 0, ConvGeomF{1, 1, Ks}, nullptr};
  if (int e = launch_f32<false>(a, (hipStream_t)stream)) return e;
  const long long n4 = (long long)M * (N >> 2);
  hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace, splits,
                     (long long)M, N, bias, relu, C, ldc);
  return (int)hipGetLastError();
}

extern "C" int combo_gemm_nt_batched_f32(const float* A, long long lda, long long sA, const float* B, long long ldb, long long sB,
                                         float* C, long long ldc, long long sC, int M, int N, int K, int batch, int relu,
                                         combo_stream_t stream) {
  if (!args_ok(A, lda, B, ldb, C, ldc, M, N, K, batch) || sA % 4 != 0 || sB % 4 != 0) return COMBO_EINVAL;
  F32Args a{A, lda, B, ldb, nullptr, C, ldc, M, N, K, relu, (int)(((M - 1LL) * ldc + N) * 4), batch, vec_ok(C, ldc, sC, N, nullptr),
            dbg_bits(), sA, sB, sC, 0, 0, 0, 0, ConvGeomF{1, 1, K}, nullptr};
  return launch_f32<false>(a, (hipStream_t)stream);
}

/* The full-resolution mask logits of ALL prediction heads in one launch (transformer_decoder.py:498-500, x 10):
 * out[h, b] = mask_embed[h, b] [Q, C] . mask_features[b] [HW, C]^T, exact fp32.  mask_embed [heads, B, Q, C], mask_features
 * [B, HW, C] token-major, out [heads, B, Q, HW].  The problems are enumerated frame-major (the `heads` problems of a frame are
 * neighbours: they share the frame's mask features in the L2 of one XCD).  The decoder's attention masks do not need these
 * logits (csrc/maskbits.hip), so they are computed after the layer loop, for the losses. */
extern "C" int combo_mask_logits_all_f32(const float* mask_embed, const float* mask_features, float* out, int heads, int B, int Q,
                                         int HW, int C, combo_stream_t stream) {
  if (heads <= 0 || B <= 0 || !args_ok(mask_embed, C, mask_features, C, out, HW, Q, HW, C, heads * B) ||
      (long long)heads * B > 0x7fffffffLL)
    return COMBO_EINVAL;
  F32Args a{mask_embed, C, mask_features, C, nullptr, out, HW, Q, HW, C, 0, (int)(((Q - 1LL) * HW + HW) * 4), heads * B,
            vec_ok(out, HW, (long long)Q * HW, HW, nullptr), dbg_bits(),
            (long long)B * Q * C, 0, (long long)B * Q * HW, heads, (long long)Q * C, (long long)HW * C, (long long)Q * HW,
            ConvGeomF{1, 1, C}, nullptr};
  return launch_f32<false>(a, (hipStream_t)stream);
}

extern "C" int combo_conv3x3_nhwc_f32(const float* X, long long ldx, const float* Wm, const float* bias, float* Y, long long ldy,
                                      int B, int H, int W, int Cin, int Cout, int relu, combo_stream_t stream) {
  const long long M = (long long)B * H * W;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % kBK != 0 || M > 0x7fffffffLL / 4 ||
      !args_ok(X, ldx, Wm, 9LL * Cin, Y, ldy, M, Cout, 9 * Cin, 1) || (bias && Cout > kMaxBiasN))
    return COMBO_EINVAL;
  F32Args a{X, ldx, Wm, 9LL * Cin, bias, Y, ldy, (int)M, Cout, 9 * Cin, relu, (int)(((M - 1) * ldy + Cout) * 4), 1,
            vec_ok(Y, ldy, 0, Cout, bias), dbg_bits(), 0, 0, 0, 0, 0, 0, 0, ConvGeomF{H, W, Cin}, nullptr};
  return launch_f32<true>(a, (hipStream_t)stream);
}
