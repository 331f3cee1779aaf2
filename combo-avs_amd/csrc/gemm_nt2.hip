// gemm_nt v2: C[M,N] = A[M,K] . B[N,K]^T (+ bias) (+ ReLU), fp32 accuracy on the bf16 matrix cores (3-way split), and the
// implicit-GEMM 3x3 convolution built on it (CONV = true, see gemm_nt.hip for the addressing).
//
// Every change against gemm_nt.hip (v1) answers one line of v1's ablation on MI355X (tools/abl_nt.py, 41160 x 256 -> 1024:
// 141 us = 18 launch/loop + 23 LDS-DMA + 5 LDS reads + 5 split + 43 MFMA + 49 epilogue, i.e. NOTHING overlapped):
//   * persistent workgroups (grid = 2 per CU) walking the tile list: no workgroup launch / teardown between tiles and the
//     first stages of the NEXT tile are already in flight while the epilogue of the current one is stored (an optional
//     half-tile start stagger of the second workgroup per CU measured neutral to -3 us: not kept)
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
#include <type_traits>

#include "combo_common.h"
#include "gemm_nt3.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float f4v __attribute__((ext_vector_type(4)));

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
    h[t] = pack_rne(v[2 * t], v[2 * t + 1]);  // hi = rne_bf16(x): the dropped lo.lo term is <= 2^-16 |x.w| and unbiased
    l[t] = pack_rne(v[2 * t] - __uint_as_float(h[t] << 16), v[2 * t + 1] - __uint_as_float(h[t] & 0xffff0000u));
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

constexpr int kBK = 16;

__device__ __attribute__((aligned(64))) float g_zero_row2[16];  // zero-initialised: the source of padded taps

struct ConvGeom2 {
  int H, W, Cin;
};

// Tile configuration: WM x WN waves (always 4), each wave owns TI x TJ MFMA tiles of 32 x 32, ST ring stages of BK = 16.
//   "wide"   4 x 1 waves, 2 x 4 tiles: 256 x 128 block, 3 stages of 24 KiB, two workgroups per CU (the big layers)
//   "mid"    2 x 2 waves, 2 x 2 tiles: 128 x 128 block, 3 stages of 16 KiB (M = 100 queries per frame pad to 128, not 256)
//   "skinny" 2 x 2 waves, 1 x 1 tiles:  64 x  64 block, 16 stages of 8 KiB = a whole K = 256 panel in flight, one
//            workgroup per CU: the decoder's M = BT*100-token layers are 32 "wide" tiles (1/8 of the chip) but 252 skinny
//            ones, and with K = 256 every byte of the tile is requested before the first MFMA waits.  Measured 11.1 us
//            at 4000 x 256 -> 256 (hipBLASLt 13 us inside the step, v1/wide 32 us); opt-in from Python
//            (COMBO_NT2_SMALL_MIN_ROWS), because the extra weight pre-split launch eats the 2 us: the small decoder
//            layers sit at the ~10 us launch + latency floor either way
template <int WM_, int WN_, int TI_, int TJ_, int ST_>
struct NtCfg {
  static constexpr int WM = WM_, WN = WN_, TI = TI_, TJ = TJ_, ST = ST_;
  static constexpr int BM = WM * TI * 32, BN = WN * TJ * 32;
  static constexpr int A_BYTES = BM * kBK * 4, B_BYTES = BN * kBK * 4, STAGE = A_BYTES + B_BYTES;
  static constexpr int A_PIECES = A_BYTES / 1024, PIECES = STAGE / 1024, PPW = PIECES / 4, APW = A_PIECES / 4;
  static constexpr int LDS = ST * STAGE;
  static_assert(WM * WN == 4 && PIECES % 4 == 0 && A_PIECES % 4 == 0, "4 waves share the DMA pieces evenly");
  static_assert((ST - 1) * PPW <= 63, "vmcnt is a 6-bit counter");
};
typedef NtCfg<4, 1, 2, 4, 3> NtWide;
typedef NtCfg<2, 2, 2, 2, 3> NtMid;      // 128 x 128 block: batched problems with ~100 rows per batch (the mask-logit contraction)
typedef NtCfg<2, 2, 1, 1, 16> NtSkinny;

template <int PPW, int ST>
__device__ __forceinline__ void wait_younger(int younger) {  // s_waitcnt vmcnt(younger * PPW): the immediate must be static
  if constexpr (ST <= 3) {  // at most one younger stage in flight
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

template <bool CONV, typename Cfg>
__global__ void __launch_bounds__(256, Cfg::LDS <= 80 * 1024 ? 2 : 1)
gemm_nt2_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ Bimg, long long ldb,
                const float* __restrict__ bias, float* __restrict__ C, long long ldc, int M, int N, int K, int relu,
                int stagger, ConvGeom2 cg, int c_bytes, int batch, long long sA, long long sB, long long sC,
                const float* __restrict__ mask, unsigned long long* ts, int products, int dbg) {
  // dbg: ablation bits (COMBO_NT2_DBG, tools/bench_nt2.py): 1 no DMA, 2 no LDS reads, 4 no barrier, 8 no stores, 16 no split, 32 no MFMA
  combo_ts_begin(ts);
  // products == 1: ONE bf16 product per multiply-add (hi . hi: plain bf16 inputs, fp32 accumulate) - the head's bf16 throughput
  // mode (forward GEMMs only); 3: the fp32-accurate split (hi . hi + hi . lo + lo . hi)
  // mask != nullptr (same shape / pitch as C): C = mask > 0 ? value : 0 - the ReLU backward of the layer whose OUTPUT was
  // the A operand's producer, folded into the input-gradient GEMM (dH = (dY . W2) o [H > 0] of an FFN)
  // batch > 1: `batch` independent problems of the same shape, operand b at A + b*sA, Bimg + b*sB, C + b*sC (elements)
  constexpr int BM = Cfg::BM, BN = Cfg::BN, TI = Cfg::TI, TJ = Cfg::TJ, ST = Cfg::ST, PPW = Cfg::PPW, APW = Cfg::APW;
  constexpr int A_BYTES = Cfg::A_BYTES, STAGE = Cfg::STAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int n_tiles = (N + BN - 1) / BN;
  const int tpb = ((M + BM - 1) / BM) * n_tiles;  // tiles per batch entry
  const int tiles = tpb * batch;
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
  unsigned tap_ok[APW];  // CONV: this lane's A rows -> 9-bit masks of the taps inside the map
#pragma unroll
  for (int u = 0; u < APW; ++u) tap_ok[u] = 0u;
  // per-piece source pointers of the open tile (row clamp, row pitch and chunk swizzle folded in once per tile): a stage
  // only adds its k offset - the per-stage 64-bit multiply-adds were ~25 of the loop's 82 VALU instructions
  const float* pa[APW];
  const float* pb[PPW - APW];
  auto open_tile = [&]() {
    const int bi = i_tile / tpb, rem = i_tile - bi * tpb;
    const float* i_A = A + bi * sA;
    const float* i_B = Bimg + bi * sB;
    i_m_blk = (rem / n_tiles) * BM;
    i_n_blk = (rem % n_tiles) * BN;
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
    if (!(dbg & 1)) {
    // CONV: the stage's tap shifts the token rows by a wave-uniform element offset
    const long long a_off = CONV ? (long long)((i_tap / 3 - 1) * cg.W + (i_tap % 3 - 1)) * lda + i_cin0 : (long long)k0;
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
      const int q = wave + 4 * u;  // wave-uniform piece (1 KiB = 16 rows x 64 B); q < A_PIECES: A rows, else B rows
      if (u < APW) {
        const float* src = pa[u < APW ? u : 0] + a_off;
        if (CONV) {
          const int c = p_chunk ^ (((q * 16 + p_row) >> 2) & 3);
          if (!((tap_ok[u < APW ? u : 0] >> i_tap) & 1u)) src = g_zero_row2 + c * 4;
        }
        glds16(src, st + q * 1024);
      } else {
        glds16(pb[u >= APW ? u - APW : 0] + k0, st + A_BYTES + (q - Cfg::A_PIECES) * 1024);
      }
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
  for (int p = 0; p < ST - 1; ++p) issue_next();

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
    // buffer descriptor over this batch entry's C for the epilogue stores (built from kernel arguments and the
    // workgroup's tile index: wave-uniform); rows >= M fall outside `c_bytes` and are dropped by the range check
    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(C + bi * sC, 0, c_bytes, 0x00020000);
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    for (int s = 0; s < nst; ++s) {
      // the stage to consume has landed once every OLDER vector-memory operation of this wave is done.  Inside a tile
      // only ring loads are pending (counted wait: the younger stages stay in flight); the first stage of a later tile
      // also drains the epilogue stores of the previous tile (stores and loads share vmcnt and may retire out of order).
      if (s == 0 && tile != w) wait_vm<0>();
      else wait_younger<PPW, ST>(issued - consumed - 1);
      if (!(dbg & 4)) __builtin_amdgcn_s_barrier();  // everybody's pieces landed; everybody finished reading the slot refilled below
      issue_next();
      const unsigned so = (unsigned)(c_slot * STAGE);
      f4v ra[TI][2], rb[TJ][2];
      if (dbg & 2) {
#pragma unroll
        for (int i = 0; i < TI; ++i) ra[i][0] = ra[i][1] = f4v{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int j = 0; j < TJ; ++j) rb[j][0] = rb[j][1] = f4v{1.f, 1.f, 1.f, 1.f};
      } else {
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
      }
      bf16x8 bh[TJ], bl[TJ], ah[TI], al[TI];
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        bh[j] = __builtin_bit_cast(bf16x8, rb[j][0]);
        bl[j] = __builtin_bit_cast(bf16x8, rb[j][1]);
      }
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        if (dbg & 16) { ah[i] = __builtin_bit_cast(bf16x8, ra[i][0]); al[i] = __builtin_bit_cast(bf16x8, ra[i][1]); }
        else split8(ra[i][0], ra[i][1], ah[i], al[i]);
      }
      if (dbg & 32) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) asm volatile("" ::"v"(ah[i]), "v"(al[i]), "v"(bh[j]), "v"(bl[j]));  // (operands stay live, accumulators untouched)
      } else {
      if (products == 3) {  // wave-uniform
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
      ++consumed;
      c_slot = c_slot == ST - 1 ? 0 : c_slot + 1;
    }

    // epilogue: D tile = 32 tokens x 32 n; lane holds n = lane & 31 and tokens (e&3) + 8*(e>>2) + 4*(lane>>5).  The
    // stores drain while the next tile's first stages (already in flight) land.  They are BUFFER stores: one descriptor over
    // C (wave-uniform), a 32-bit byte offset per lane that walks the rows by additions of ldc / 5*ldc, and the hardware
    // range check drops the rows >= M of the last token tile.  (The first version's per-store 64-bit multiply-adds, row
    // tests and exec masking were ~2 000 VALU instructions per tile: 60 % of the kernel's VALU count by PMC.)
    auto epilogue = [&](auto relu_tag, auto mask_tag) {
      constexpr bool RELU = decltype(relu_tag)::value, MASK = decltype(mask_tag)::value;
      const unsigned uld = (unsigned)ldc * 4u, uld5 = 5u * uld;
      const __amdgpu_buffer_rsrc_t m_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MASK ? mask + bi * sC : C), 0, c_bytes, 0x00020000);
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int n = n_blk + (wn * TJ + j) * 32 + m;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
        unsigned off = ((unsigned)(m_blk + wm * TI * 32 + 4 * g) * (unsigned)ldc + (unsigned)n) * 4u;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float v = acc[i][j][e] + bv;
            if (RELU) v = fmaxf(v, 0.f);
            if (MASK) v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(m_rsrc, off, 0, 0)) > 0.f ? v : 0.f;
            if (dbg & 8) asm volatile("" ::"v"(v));
            else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), c_rsrc, off, 0, 0);
            off += (e & 3) == 3 ? uld5 : uld;  // rows 0-3, 8-11, 16-19, 24-27 (+4g); the next i starts 32 rows on
          }
      }
    };
    if (mask) epilogue(std::false_type{}, std::true_type{});
    else if (relu) epilogue(std::true_type{}, std::false_type{});
    else epilogue(std::false_type{}, std::false_type{});
  }
  combo_ts_end(ts);
}

// Weight image for gemm_nt2: element (n, k) = src[n*ld_row + k*ld_col]; per 8 consecutive k a 16-B group of bf16 `hi`
// = rne(x) followed by a 16-B group of bf16 `lo` = rne(x - hi): 4 bytes per element, row stride K floats.
__global__ void __launch_bounds__(256)
presplit_kernel(const float* __restrict__ src, long long ld_row, long long ld_col, int N, int K, uint4* __restrict__ img,
                long long batch_stride) {
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
  for (int i = 0; i < 4; ++i) {
    h[i] = pack_rne(v[2 * i], v[2 * i + 1]);
    l[i] = pack_rne(v[2 * i] - __uint_as_float(h[i] << 16), v[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u));
  }
  const long long o = ((long long)n * kg + g8) * 2;
  img[o] = make_uint4(h[0], h[1], h[2], h[3]);
  img[o + 1] = make_uint4(l[0], l[1], l[2], l[3]);
}

// Grouped form: every weight whose input-gradient GEMM the backward pass will run, split by ONE launch (per 64 problems)
// instead of one ~5 us launch per weight (148 per training step).  img_ld: floats per image row (>= K): several sources may
// fill k ranges of one image (the two weights of a column-concatenated layer).
constexpr int kMaxSplitGroup = 64;
struct SplitGroupArgs {
  int count;
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
  int n, g8;
  if (pr.ld_col == 1) { n = (int)(t / kg); g8 = (int)(t - (long long)n * kg); }
  else { g8 = (int)(t / pr.N); n = (int)(t - (long long)g8 * pr.N); }
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = pr.src[(long long)n * pr.ld_row + (long long)(g8 * 8 + i) * pr.ld_col];
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = pack_rne(v[2 * i], v[2 * i + 1]);
    l[i] = pack_rne(v[2 * i] - __uint_as_float(h[i] << 16), v[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u));
  }
  uint4* img = reinterpret_cast<uint4*>(pr.img) + (long long)n * (pr.img_ld >> 2) + g8 * 2;
  img[0] = make_uint4(h[0], h[1], h[2], h[3]);
  img[1] = make_uint4(l[0], l[1], l[2], l[3]);
}

struct NtBatch {
  int batch;
  long long sA, sB, sC;
};

int nt2_dbg_bits() {
  static const int d = [] { const char* e = getenv("COMBO_NT2_DBG"); return e ? atoi(e) : 0; }();
  return d;
}

bool use_nt3() {  // COMBO_DX_KERNEL=2: the round-3 kernel of this file (A/B during the round-4 rewrite, csrc/gemm_nt3.hip)
  static const bool on = [] { const char* e = getenv("COMBO_DX_KERNEL"); return !(e && atoi(e) == 2); }();
  return on;
}

int g_products = 3;  // combo_gemm_nt2_products: bf16 products per multiply-add of the launches that follow (host state, read at launch)

template <bool CONV, typename Cfg>
int launch_nt2_cfg(const float* A, long long lda, const float* Bimg, const float* bias, float* C, long long ldc, long long M,
                   int N, int K, int relu, ConvGeom2 cg, int n_cu, combo_stream_t stream, NtBatch nb = NtBatch{1, 0, 0, 0},
                   const float* mask = nullptr) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt2_kernel<CONV, Cfg>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const long long tiles = ((M + Cfg::BM - 1) / Cfg::BM) * ((N + Cfg::BN - 1) / Cfg::BN) * nb.batch;
  if (tiles > 0x7fffffffLL) return COMBO_EINVAL;
  const long long slots = (Cfg::LDS <= 80 * 1024 ? 2LL : 1LL) * n_cu;  // resident workgroups
  const int grid = (int)(tiles < slots ? tiles : slots);
  const int stagger = 0;  // (a half-tile start stagger of the second workgroup per CU measured neutral to -3 us: not kept)
  // extent of C in bytes for the epilogue's buffer descriptor (32-bit offsets)
  const long long c_bytes = ((M - 1) * ldc + N) * 4;
  if (c_bytes >= 0x7fffffffLL) return COMBO_EINVAL;
  hipLaunchKernelGGL((gemm_nt2_kernel<CONV, Cfg>), dim3((unsigned)grid), dim3(256), Cfg::LDS, (hipStream_t)stream, A, lda, Bimg,
                     (long long)K, bias, C, ldc, (int)M, N, K, relu, stagger, cg, (int)c_bytes, nb.batch, nb.sA, nb.sB, nb.sC, mask,
                     combo_timing_next_slot(g_products == 3 ? COMBO_TS_GEMM_X3 : COMBO_TS_GEMM_BF16, 2.0 * M * N * K * nb.batch,
                                            4.0 * nb.batch * ((double)M * (CONV ? K / 9 : K) + (double)N * K + (double)M * N)),
                     g_products, nt2_dbg_bits());
  return (int)hipGetLastError();
}

template <bool CONV>
int launch_nt2(const float* A, long long lda, const float* Bimg, const float* bias, float* C, long long ldc, long long M, int N,
               int K, int relu, ConvGeom2 cg, combo_stream_t stream, const float* mask = nullptr) {
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return cus > 0 ? cus : 256;
  }();
  static const int skinny_mode = [] { const char* e = getenv("COMBO_NT2_SKINNY"); return e ? atoi(e) : 1; }();  // 0 never, 2 always
  const long long wide_tiles = ((M + NtWide::BM - 1) / NtWide::BM) * ((N + NtWide::BN - 1) / NtWide::BN);
  // skinny tiles when the wide ones would leave most of the chip idle (the decoder's M = BT*100-token layers)
  const bool skinny = skinny_mode == 2 || (skinny_mode == 1 && wide_tiles * 2 <= n_cu);
  if (skinny) return launch_nt2_cfg<CONV, NtSkinny>(A, lda, Bimg, bias, C, ldc, M, N, K, relu, cg, n_cu, stream, NtBatch{1, 0, 0, 0}, mask);
  return launch_nt2_cfg<CONV, NtWide>(A, lda, Bimg, bias, C, ldc, M, N, K, relu, cg, n_cu, stream, NtBatch{1, 0, 0, 0}, mask);
}

}  // namespace

/* bf16 products per fp32 multiply-add of the combo_gemm_nt_x3_* / combo_conv3x3_nhwc_x3_* launches that FOLLOW (3 = the
 * fp32-accurate split, the default; 1 = plain bf16 inputs with fp32 accumulation: the head's bf16 throughput mode, forward GEMMs
 * only).  Returns the previous value.  Host-side state, read when a launch is issued (a captured graph keeps what was set). */
extern "C" int combo_gemm_nt2_products(int products) {
  const int prev = g_products;
  if (products == 1 || products == 3) g_products = products;
  return prev;
}

extern "C" int combo_presplit_bf16x2_f32(const float* src, long long ld_row, long long ld_col, int N, int K, float* img,
                                         combo_stream_t stream) {
  if (!src || !img || N <= 0 || K <= 0 || K % 8 != 0 || ((uintptr_t)img & 15)) return COMBO_EINVAL;
  const long long threads = (long long)N * (K / 8);
  hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld_row,
                     ld_col, N, K, reinterpret_cast<uint4*>(img), 0LL);
  return (int)hipGetLastError();
}

extern "C" int combo_presplit_bf16x2_grouped_f32(const combo_presplit_problem* problems, int count, combo_stream_t stream) {
  if (!problems || count <= 0) return COMBO_EINVAL;
  for (int base = 0; base < count; base += kMaxSplitGroup) {
    SplitGroupArgs a;
    a.count = count - base < kMaxSplitGroup ? count - base : kMaxSplitGroup;
    long long threads = 0;
    for (int i = 0; i < a.count; ++i) {
      const combo_presplit_problem& pr = problems[base + i];
      if (!pr.src || !pr.img || pr.N <= 0 || pr.K <= 0 || pr.K % 8 != 0 || pr.img_ld < pr.K || pr.img_ld % 8 != 0 ||
          ((uintptr_t)pr.img & 31))
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
                     src, ld_row, ld_col, N, K, reinterpret_cast<uint4*>(img), batch_stride);
  return (int)hipGetLastError();
}

// `batch` independent GEMMs of one shape (the mask-logit contraction mask_embed @ pixel_embed^T per frame and its input
// gradient): C_b[M,N] = A_b[M,K] . B_b[N,K]^T, operands at base + b * stride (elements).  128 x 128 tiles when M pads
// better to 128 than to 256.
extern "C" int combo_gemm_nt_x3_pre_batched_f32(const float* A, long long lda, long long sA, const float* Bimg, long long sB,
                                                float* C, long long ldc, long long sC, int M, int N, int K, int batch, int relu,
                                                combo_stream_t stream) {
  if (!A || !Bimg || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || K % kBK != 0 || lda % 4 != 0 || sA % 4 != 0 ||
      sB % 4 != 0 || ((uintptr_t)A & 15) || ((uintptr_t)Bimg & 15))
    return COMBO_EINVAL;
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return cus > 0 ? cus : 256;
  }();
  const int pad256 = (M + 255) / 256 * 256, pad128 = (M + 127) / 128 * 128;
  if (use_nt3())
    return combo_nt3_launch(A, lda, Bimg, K, nullptr, nullptr, C, ldc, M, N, K, relu, g_products, batch, sA, sB, sC, nullptr,
                            pad128 < pad256 ? 2 : 1, stream);
  const NtBatch nb{batch, sA, sB, sC};
  if (pad128 < pad256)
    return launch_nt2_cfg<false, NtMid>(A, lda, Bimg, nullptr, C, ldc, M, N, K, relu, ConvGeom2{1, 1, K}, n_cu, stream, nb);
  return launch_nt2_cfg<false, NtWide>(A, lda, Bimg, nullptr, C, ldc, M, N, K, relu, ConvGeom2{1, 1, K}, n_cu, stream, nb);
}

extern "C" int combo_gemm_nt_x3_pre_f32(const float* A, long long lda, const float* Bimg, const float* bias, float* C,
                                        long long ldc, int M, int N, int K, int relu, combo_stream_t stream) {
  if (!A || !Bimg || !C || M <= 0 || N <= 0 || K <= 0 || K % kBK != 0 || lda % 4 != 0 || ((uintptr_t)A & 15) ||
      ((uintptr_t)Bimg & 15))
    return COMBO_EINVAL;
  if (use_nt3() && !(bias && N > 2048))
    return combo_nt3_launch(A, lda, Bimg, K, bias, nullptr, C, ldc, M, N, K, relu, g_products, 1, 0, 0, 0, nullptr, 0, stream);
  return launch_nt2<false>(A, lda, Bimg, bias, C, ldc, M, N, K, relu, ConvGeom2{1, 1, K}, stream);
}

extern "C" int combo_gemm_nt_x3_pre_masked_f32(const float* A, long long lda, const float* Bimg, const float* mask, float* C,
                                               long long ldc, int M, int N, int K, combo_stream_t stream) {
  if (!A || !Bimg || !C || !mask || M <= 0 || N <= 0 || K <= 0 || K % kBK != 0 || lda % 4 != 0 || ((uintptr_t)A & 15) ||
      ((uintptr_t)Bimg & 15))
    return COMBO_EINVAL;
  if (use_nt3())
    return combo_nt3_launch(A, lda, Bimg, K, nullptr, mask, C, ldc, M, N, K, 0, g_products, 1, 0, 0, 0, nullptr, 0, stream);
  return launch_nt2<false>(A, lda, Bimg, nullptr, C, ldc, M, N, K, 0, ConvGeom2{1, 1, K}, stream, mask);
}

extern "C" int combo_conv3x3_nhwc_x3_pre_f32(const float* X, long long ldx, const float* Wimg, const float* bias, float* Y,
                                             long long ldy, int B, int H, int W, int Cin, int Cout, int relu,
                                             combo_stream_t stream) {
  const long long M = (long long)B * H * W;
  if (!X || !Wimg || !Y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % kBK != 0 || ldx % 4 != 0 ||
      ((uintptr_t)X & 15) || ((uintptr_t)Wimg & 15) || M > 0x7fffffffLL / 4)
    return COMBO_EINVAL;
  if (use_nt3() && !(bias && Cout > 2048)) {
    const combo_nt3_conv cg{H, W, Cin};
    return combo_nt3_launch(X, ldx, Wimg, 9LL * Cin, bias, nullptr, Y, ldy, M, Cout, 9 * Cin, relu, g_products, 1, 0, 0, 0, &cg, 0, stream);
  }
  return launch_nt2<true>(X, ldx, Wimg, bias, Y, ldy, M, Cout, 9 * Cin, relu, ConvGeom2{H, W, Cin}, stream);
}
