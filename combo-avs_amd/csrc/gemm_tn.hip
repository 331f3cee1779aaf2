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

__device__ __forceinline__ unsigned pack_rne(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

struct Frag {
  bf16x8 hi, lo;
};

// Four packs per asm statement, each ending in `s_nop 1`: the fragments feed matrix instructions, and hipcc pads no hazard whose
// producer sits inside an asm string (a VALU-written VGPR needs 2 wait states before an MFMA reads it as A / B; round 5: the same
// unpadded pattern in csrc/sra_attention.hip gave stale operands on ~4 % of its tiles, and a scan of this file's ISA showed 7
// v_cvt_pk -> v_mfma pairs closer than that).
__device__ __forceinline__ void pack4(const float (&x)[8], unsigned (&r)[4]) {
  asm("v_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %6, %7\n\tv_cvt_pk_bf16_f32 %2, %8, %9\n\tv_cvt_pk_bf16_f32 %3, %10, %11\n\ts_nop 1"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
}

__device__ __forceinline__ Frag make_frag(const float (&v)[8]) {
  Frag f;
  unsigned h[4], l[4];
  pack4(v, h);  // hi = rne_bf16(x): the dropped lo.lo term is <= 2^-16 |x.w| and unbiased
  float res[8];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    res[2 * t] = v[2 * t] - __uint_as_float(h[t] << 16);
    res[2 * t + 1] = v[2 * t + 1] - __uint_as_float(h[t] & 0xffff0000u);
  }
  pack4(res, l);
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


// -------------------------------------------------------------------------------------------------------------------
// v2: LDS-DMA ring + interleaved wave tiles (for N % 4 == 0, K % 4 == 0, 16-byte aligned operands).
//
// The direct-from-global kernel above is bound by vector-memory ISSUE: 32 dword loads per wave for 12 MFMAs, one
// k-step of prefetch.  Here a workgroup streams [16 tokens] x [BN + BK columns] stages through a 3-deep LDS ring with
// global_load_lds_dwordx4 (no VGPR round trip, two stages in flight across a raw s_barrier, counted s_waitcnt vmcnt),
// and every wave owns a 128(n) x 64(k) tile whose n/k indices are INTERLEAVED: MFMA tile i of the A side holds
// n = n0 + 4*row + i, so ONE ds_read_b128 of four consecutive n for a token feeds the fragments of four MFMA tiles (and
// one ds_read_b64 two B tiles): 16 conflict-free LDS reads for 24 MFMAs per k-step, and the token-major operands never
// need a transpose.  WN = waves along n: block tile 256 x 128 (WN = 2) or 128 x 256 (WN = 1).
// -------------------------------------------------------------------------------------------------------------------
constexpr int kTS = 16;      // tokens (reduction rows) per stage

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

// LDS reads as inline asm: hipcc drains every outstanding LDS-DMA (s_waitcnt vmcnt(0)) in front of a ds_read it can
// see, which would serialise the ring; ordering is done by hand (counted vmcnt -> s_barrier -> these reads).
__device__ __forceinline__ f4v lds_read128(unsigned addr) {
  f4v r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
__device__ __forceinline__ f2v lds_read64(unsigned addr) {
  f2v r;
  asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void glds16(const float* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// CONV (weight gradient of the 3x3 / stride 1 / pad 1 FPN output convolution, see gemm_nt.hip): K = 9*Cin ordered
// (tap, cin) and a k tile lies inside one tap (Cin % BK == 0), so the X rows of a stage are the token rows shifted by the
// tap, or a row of zeros where the tap leaves the map.  Token -> (y, x) by multiply-high with host-made reciprocals.
__device__ __attribute__((aligned(64))) float g_tn_zero_row[256];

// (round 5) generalised: H, W = the OUTPUT map (the rows of dY), Hin, Win = the input map (the rows of X), stride 1 or 2, kernel
// size 1 or 3 (padding ksize / 2); a k tile may span several taps (Cin = 64: two taps per 128 columns) - every lane keeps the
// (tap, channel) of ITS 16-byte column group (Cin % 4 == 0).
struct TnConvGeom {
  int H, W, Cin;
  unsigned magicW, magicH;  // ceil(2^32 / W), ceil(2^32 / H): the quotient is exact while tokens * max(H, W) < 2^32 (host check)
  int Hin, Win, stride, ksize;
};

template <int WN, int kStages, bool CONV = false>
__device__ __forceinline__ void
gemm_tn_glds_body(const float* __restrict__ dY, long long ldy, const float* __restrict__ X, long long ldx,
                  float* __restrict__ out, float* __restrict__ db_part, int M, int N, int K, int mchunk, int bx, int by,
                  int bz, TnConvGeom cg = TnConvGeom{1, 1, 1, 0u, 0u, 1, 1, 1, 3}) {
  constexpr int WK = 4 / WN, BN = 128 * WN, BK = 64 * WK;
  constexpr int A_BYTES = kTS * BN * 4, B_BYTES = kTS * BK * 4, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PIECES = A_BYTES / 1024, B_PIECES = B_BYTES / 1024, PIECES = A_PIECES + B_PIECES;  // 1-KiB DMA pieces
  constexpr int PPW = PIECES / 4;  // pieces per wave per stage (24 KiB / 4 waves = 6)
  static_assert(PIECES % 4 == 0, "pieces must divide over the 4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave / WK, wk = wave % WK;
  const int n_blk = by * BN, k_blk = bx * BK;
  const int mbeg = bz * mchunk, mend = min(M, mbeg + mchunk);
  const int nst = (mend - mbeg + kTS - 1) / kTS;
  const int col = lane & 31, kg = lane >> 5;

  int l_dy = 0, l_dx = 0, l_ci = 0;  // CONV: tap offset and input channel of this lane's X column group (constant over the stages)
  if (CONV) {
    const int cl = min(BK == 128 ? k_blk + (lane & 31) * 4 : k_blk + lane * 4, K - 4);
    const int tap = cl / cg.Cin, pad = cg.ksize >> 1;
    l_ci = cl - tap * cg.Cin;
    l_dy = tap / cg.ksize - pad;
    l_dx = tap % cg.ksize - pad;
  }
  // ---- per-lane DMA source columns (constant over the stages), clamped inside the matrix ----
  // piece q of a stage (1 KiB = one wave instruction): q < A_PIECES -> dY rows, else X rows
  auto issue = [&](int s) {
    char* st = smem + (s % kStages) * STAGE;
    const int m0 = mbeg + s * kTS;
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
      const int q = wave + 4 * u;  // wave-uniform
      if (q < A_PIECES) {
        int t, c;
        if (BN == 256) { t = q; c = n_blk + lane * 4; }                          // one 1-KiB row per piece
        else { t = q * 2 + (lane >> 5); c = n_blk + (lane & 31) * 4; }           // two 512-B rows per piece
        c = min(c, N - 4);
        const int m = min(m0 + t, M - 1);
        glds16(dY + (long long)m * ldy + c, st + q * 1024);
      } else {
        const int qb = q - A_PIECES;
        int t, c;
        if (BK == 128) { t = qb * 2 + (lane >> 5); c = k_blk + (lane & 31) * 4; }
        else { t = qb; c = k_blk + lane * 4; }
        c = min(c, K - 4);
        const int m = min(m0 + t, M - 1);
        if (CONV) {
          const unsigned row = __umulhi((unsigned)m, cg.magicW);          // m / W            (output token m = (frame, y, x))
          const int x = m - (int)row * cg.W;
          const unsigned fr = __umulhi(row, cg.magicH);                   // frame
          const int y = (int)row - (int)fr * cg.H;
          const int yi = y * cg.stride + l_dy, xi = x * cg.stride + l_dx;  // the input pixel under this lane's tap
          const bool ok = (unsigned)yi < (unsigned)cg.Hin && (unsigned)xi < (unsigned)cg.Win;
          const float* src = ok ? X + ((long long)((int)fr * cg.Hin + yi) * cg.Win + xi) * ldx + l_ci : g_tn_zero_row + (l_ci & 255);
          glds16(src, st + A_BYTES + qb * 1024);
        } else {
          glds16(X + (long long)m * ldx + c, st + A_BYTES + qb * 1024);
        }
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
  const bool do_db = db_part != nullptr && bx == 0 && wk == 0;
  float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);

#pragma unroll
  for (int p = 0; p < kStages - 1; ++p)
    if (p < nst) issue(p);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned a_off = lds0 + (unsigned)((kg * 8 * BN + wn * 128 + col * 4) * 4);
  const unsigned b_off = lds0 + (unsigned)(A_BYTES + (kg * 8 * BK + wk * 64 + col * 2) * 4);
  for (int s = 0; s < nst; ++s) {
    // stage s has landed once MY pieces of it are done (counted: up to kStages-2 younger stages stay in flight) and
    // everybody passed the barrier; the barrier also says everybody finished reading stage s-1, whose slot the stage
    // issued below reuses
    const int younger = min(kStages - 2, nst - 1 - s);
    if (younger >= 3) wait_vm<3 * PPW>();
    else if (younger == 2) wait_vm<2 * PPW>();
    else if (younger == 1) wait_vm<PPW>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (s + kStages - 1 < nst) issue(s + kStages - 1);
    const unsigned so = (unsigned)((s % kStages) * STAGE);
    f4v ra[8];
    f2v rb[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      ra[t] = lds_read128(a_off + so + t * (BN * 4));
      rb[t] = lds_read64(b_off + so + t * (BK * 4));
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]),
                   "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]), "+v"(rb[5]), "+v"(rb[6]), "+v"(rb[7])
                 :
                 : "memory");
    float4 va[8];
    float2 vb[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      va[t] = make_float4(ra[t].x, ra[t].y, ra[t].z, ra[t].w);
      vb[t] = make_float2(rb[t].x, rb[t].y);
    }
    if (mbeg + (s + 1) * kTS > mend) {  // ragged last stage: rows beyond mend contribute nothing (zero the A side)
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (mbeg + s * kTS + kg * 8 + t >= mend) va[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (do_db) {
#pragma unroll
      for (int t = 0; t < 8; ++t) { colsum.x += va[t].x; colsum.y += va[t].y; colsum.z += va[t].z; colsum.w += va[t].w; }
    }
    Frag fa[4], fb[2];
    {
      float v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = va[t].x;
      fa[0] = make_frag(v);
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = va[t].y;
      fa[1] = make_frag(v);
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = va[t].z;
      fa[2] = make_frag(v);
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = va[t].w;
      fa[3] = make_frag(v);
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = vb[t].x;
      fb[0] = make_frag(v);
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = vb[t].y;
      fb[1] = make_frag(v);
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
  const int n0 = n_blk + wn * 128, k0 = k_blk + wk * 64;
  if (do_db) {
    colsum.x += __shfl_xor(colsum.x, 32); colsum.y += __shfl_xor(colsum.y, 32);
    colsum.z += __shfl_xor(colsum.z, 32); colsum.w += __shfl_xor(colsum.w, 32);
    const int nr = n0 + col * 4;
    if (kg == 0 && nr < N) *reinterpret_cast<float4*>(db_part + (long long)bz * N + nr) = colsum;
  }
  float* o = out + (long long)bz * N * K;
  const int kc = k0 + col * 2;
  if (kc < K) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int nr = n0 + 4 * ((e & 3) + 8 * (e >> 2) + 4 * kg) + i;
        if (nr < N) *reinterpret_cast<float2*>(o + (long long)nr * K + kc) = make_float2(acc[i][0][e], acc[i][1][e]);
      }
  }
}

template <int WN, int kStages>
__global__ void __launch_bounds__(256, kStages <= 3 ? 2 : 1)
gemm_tn_glds_kernel(const float* __restrict__ dY, long long ldy, const float* __restrict__ X, long long ldx,
                    float* __restrict__ out, float* __restrict__ db_part, int M, int N, int K, int mchunk, int remap) {
  // the tiles of one token chunk (blockIdx.z) read the same dY / X rows: keep them on one XCD (see xcd_contiguous)
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (remap) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int l = xcd_contiguous(bx + gx * (by + gy * bz), gx * gy * (int)gridDim.z);
    bx = l % gx; by = (l / gx) % gy; bz = l / (gx * gy);
  }
  gemm_tn_glds_body<WN, kStages>(dY, ldy, X, ldx, out, db_part, M, N, K, mchunk, bx, by, bz);
}

// WN = 2: 256 (cout) x 128 (k) block tiles; WN = 1 (round 5): 128 x 256 - the 64- and 128-channel layers of res2 / res3 waste
// 3/4 and 1/2 of a 256-wide cout tile (clamped DMA columns, MFMAs on duplicates)
template <int WN>
__global__ void __launch_bounds__(256, 2)
conv3x3_wgrad_kernel(const float* __restrict__ dY, long long ldy, const float* __restrict__ X, long long ldx,
                     float* __restrict__ out, int M, int N, int K, int mchunk, int remap, TnConvGeom cg, unsigned long long* ts) {
  combo_ts_begin(ts);
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (remap) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int l = xcd_contiguous(bx + gx * (by + gy * bz), gx * gy * (int)gridDim.z);
    bx = l % gx; by = (l / gx) % gy; bz = l / (gx * gy);
  }
  gemm_tn_glds_body<WN, 3, true>(dY, ldy, X, ldx, out, nullptr, M, N, K, mchunk, bx, by, bz, cg);
  combo_ts_end(ts);
}

// Grouped launch: many independent weight-gradient problems in ONE kernel.  The decoder's dW GEMMs (M = BT*100 tokens)
// are latency-bound (~25 us each for 0.5 GFLOP, 114 of them per step) and not on the backward critical path, so
// ops/linear.py defers them and runs them together at the end of the backward pass.  The problem table travels in the
// kernel arguments (so it is frozen into a captured hipGraph together with the pointers).
constexpr int kMaxGroup = 40;
struct TnGroupArgs {
  int count;
  int remap;
  int block_start[kMaxGroup + 1];  // multiples of 8 when remap is on, so that (b - start) % 8 is the block's XCD
  int block_count[kMaxGroup];
  unsigned long long* ts;  // device-side timing slot (combo_common.h) or nullptr
  combo_gemm_tn_problem p[kMaxGroup];
};

__global__ void __launch_bounds__(256, 2)
gemm_tn_grouped_kernel(const TnGroupArgs args) {
  combo_ts_begin(args.ts);
  const int b = blockIdx.x;
  int pi = 0;
  for (int i = 1; i < args.count; ++i)
    if (b >= args.block_start[i]) pi = i;
  const combo_gemm_tn_problem& pr = args.p[pi];
  int local = b - args.block_start[pi];
  if (local < args.block_count[pi]) {  // (else: padding block)
    // the tiles of one token chunk read the same dY / X rows: keep them on one XCD (see xcd_contiguous)
    if (args.remap) local = xcd_contiguous(local, args.block_count[pi]);
    int mchunk = (pr.M + pr.splits - 1) / pr.splits;
    mchunk = (mchunk + 15) / 16 * 16;
    const long long t2 = (long long)((pr.N + 255) / 256) * ((pr.K + 127) / 128) * 256 * 128;
    const long long t1 = (long long)((pr.N + 127) / 128) * ((pr.K + 255) / 256) * 128 * 256;
    if (t2 <= t1) {
      const int tk = (pr.K + 127) / 128, tn = (pr.N + 255) / 256;
      gemm_tn_glds_body<2, 3>(pr.dY, pr.ldy, pr.X, pr.ldx, pr.partials, pr.db_partials, pr.M, pr.N, pr.K, mchunk, local % tk,
                              (local / tk) % tn, local / (tk * tn));
    } else {
      const int tk = (pr.K + 255) / 256, tn = (pr.N + 127) / 128;
      gemm_tn_glds_body<1, 3>(pr.dY, pr.ldy, pr.X, pr.ldx, pr.partials, pr.db_partials, pr.M, pr.N, pr.K, mchunk, local % tk,
                              (local / tk) % tn, local / (tk * tn));
    }
  }
  combo_ts_end(args.ts);
}

inline bool glds_ok(const float* dY, long long ldy, const float* X, long long ldx, int M, int N, int K) {
  return N % 4 == 0 && K % 4 == 0 && N >= 64 && K >= 64 && ldy % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)dY & 15) == 0 &&
         ((uintptr_t)X & 15) == 0 && M >= 256;
}

// out[i] = sum_z part[z*n + i] (i < n) and db[j] = sum_z db_part[z*nb + j] (j < nb): ONE launch finishes the split-K
// weight gradient and the fused bias gradient, in a fixed summation order (deterministic), straight into the caller's
// destination (e.g. a row block of a packed in_proj gradient).
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float* __restrict__ part, int splits, long long n, float* __restrict__ out,
                     const float* __restrict__ db_part, int nb, float* __restrict__ db) {
  const long long n4 = n >> 2;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  // eight partials in flight per thread, summed in the same fixed order: a tiny output (the class head: 3 x 256 values from
  // 157 token slices) is ONE workgroup walking the splits - one memory latency per split (281 us) without the batching
  if (i < n4) {
    float4 a = reinterpret_cast<const float4*>(part)[i];
    int z = 1;
    for (; z + 8 <= splits; z += 8) {
      float4 b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = reinterpret_cast<const float4*>(part + (long long)(z + u) * n)[i];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
    }
    for (; z < splits; ++z) {
      const float4 b = reinterpret_cast<const float4*>(part + (long long)z * n)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
  } else if (i - n4 < nb) {
    const int j = (int)(i - n4);
    float a = db_part[j];
    int z = 1;
    for (; z + 8 <= splits; z += 8) {
      float b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = db_part[(long long)(z + u) * nb + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += b[u];
    }
    for (; z < splits; ++z) a += db_part[(long long)z * nb + j];
    db[j] = a;
  }
}

// The same sum for a 3x3 convolution's weight gradient, written in the parameter's own (NCHW) order: partials are
// [splits][Cout][taps][Cin] (the implicit GEMM's K order), out is [Cout][Cin][taps].  A thread sums one float4 of the partial
// layout (4 consecutive cin of one (cout, tap): coalesced reads, the loop depth of the plain reduce) and scatters its 4 results
// `taps` floats apart - no permute copy after the reduce.  Cin % 4 == 0.
__global__ void __launch_bounds__(256)
splitk_reduce_nchw_kernel(const float* __restrict__ part, int splits, int cout, int taps, int cin, float* __restrict__ out) {
  const long long n = (long long)cout * taps * cin;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;  // float4 index into [Cout][taps][Cin]
  if (i >= (n >> 2)) return;
  float4 a = reinterpret_cast<const float4*>(part)[i];
  for (int z = 1; z < splits; ++z) {
    const float4 b = reinterpret_cast<const float4*>(part + (long long)z * n)[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  const long long e = i * 4;
  const int ci = (int)(e % cin);
  const long long ct = e / cin;  // co * taps + tap
  const int tap = (int)(ct % taps);
  const long long co = ct / taps;
  float* o = out + (co * cin + ci) * taps + tap;
  o[0] = a.x; o[taps] = a.y; o[2 * taps] = a.z; o[3 * taps] = a.w;
}

struct ReduceGroupArgs {
  int count;
  long long thread_start[kMaxGroup + 1];
  combo_reduce_problem p[kMaxGroup];
};

__global__ void __launch_bounds__(256)
splitk_reduce_grouped_kernel(const ReduceGroupArgs args) {
  const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t >= args.thread_start[args.count]) return;
  int pi = 0;
  for (int i = 1; i < args.count; ++i)
    if (t >= args.thread_start[i]) pi = i;
  const combo_reduce_problem& pr = args.p[pi];
  const long long i = t - args.thread_start[pi];
  const long long n4 = pr.n >> 2;
  if (i < n4) {
    float4 a = reinterpret_cast<const float4*>(pr.partials)[i];
    int z = 1;
    for (; z + 8 <= pr.splits; z += 8) {  // 8 slices in flight (the loop is latency-bound), added in the same fixed order
      float4 b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = reinterpret_cast<const float4*>(pr.partials + (long long)(z + u) * pr.n)[i];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
    }
    for (; z < pr.splits; ++z) {
      const float4 b = reinterpret_cast<const float4*>(pr.partials + (long long)z * pr.n)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(pr.out)[i] = a;
  } else if (i - n4 < pr.nb) {
    const int j = (int)(i - n4);
    float a = pr.db_partials[j];
    for (int z = 1; z < pr.splits; ++z) a += pr.db_partials[(long long)z * pr.nb + j];
    pr.db[j] = a;
  }
}

}  // namespace

extern "C" {

int combo_splitk_reduce_f32(const float* partials, int splits, long long n, float* out, const float* db_partials, int nb,
                            float* db, combo_stream_t stream) {
  if (!partials || !out || splits <= 0 || n <= 0 || (n & 3) || ((uintptr_t)partials & 15) || ((uintptr_t)out & 15) ||
      (nb > 0 && (!db_partials || !db)))
    return COMBO_EINVAL;
  const long long threads = (n >> 2) + (nb > 0 ? nb : 0);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     partials, splits, n, out, db_partials, nb > 0 ? nb : 0, db);
  return (int)hipGetLastError();
}

int combo_splitk_reduce_nchw_f32(const float* partials, int splits, int Cout, int taps, int Cin, float* out, combo_stream_t stream) {
  if (!partials || !out || splits <= 0 || Cout <= 0 || taps <= 0 || Cin <= 0 || Cin % 4 != 0 || ((uintptr_t)partials & 15)) return COMBO_EINVAL;
  const long long threads = (long long)Cout * taps * Cin / 4;
  hipLaunchKernelGGL(splitk_reduce_nchw_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partials,
                     splits, Cout, taps, Cin, out);
  return (int)hipGetLastError();
}

static int tn_variant(int N, int K) {  // 2: 256(n) x 128(k) block tiles, 1: 128 x 256
  const long long t2 = (long long)((N + 255) / 256) * ((K + 127) / 128) * 256 * 128;
  const long long t1 = (long long)((N + 127) / 128) * ((K + 255) / 256) * 128 * 256;
  return t2 <= t1 ? 2 : 1;  // less padded area wins
}

int combo_gemm_tn_splits(int M, int N, int K) {
  if (N % 4 == 0 && K % 4 == 0 && N >= 64 && K >= 64 && M >= 256) {
    const int v = tn_variant(N, K);
    const long long tiles = v == 2 ? (long long)((N + 255) / 256) * ((K + 127) / 128)
                                   : (long long)((N + 127) / 128) * ((K + 255) / 256);
    // measured (tools/bench_dw.py, a sweep over the split count): one workgroup per CU for outputs of <= 4 tiles (every
    // extra split costs a 128-KiB partial tile written and re-read), two per CU for larger outputs
    long long s = ((tiles <= 4 ? 256 : 512) + tiles - 1) / tiles;
    const long long maxs = (M + 127) / 128;   // at least 8 stages per split
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    return (int)s;
  }
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
  if (glds_ok(dY, ldy, X, ldx, M, N, K)) {  // (else: the direct-from-global kernel below, any alignment)
    constexpr int stage_bytes = kTS * 256 * 4 + kTS * 128 * 4;  // both variants: 24 KiB
    const int stages = 3;  // 3 stages x 2 workgroups/CU (measured against 5 stages x 1 workgroup/CU: the instances stay for tools)
    static ComboDevFlag attr;
    if (!attr.is_set()) {
      const void* fns[4] = {reinterpret_cast<const void*>(gemm_tn_glds_kernel<2, 3>), reinterpret_cast<const void*>(gemm_tn_glds_kernel<1, 3>),
                            reinterpret_cast<const void*>(gemm_tn_glds_kernel<2, 5>), reinterpret_cast<const void*>(gemm_tn_glds_kernel<1, 5>)};
      for (int i = 0; i < 4; ++i) {
        hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (i < 2 ? 3 : 5) * stage_bytes);
        if (e != hipSuccess) return (int)e;
      }
      attr.mark();
    }
    const int lds = stages * stage_bytes;
    const int remap = 1;  // XCD-contiguous tile order
    const bool v2 = tn_variant(N, K) == 2;
    const dim3 grid = v2 ? dim3((K + 127) / 128, (N + 255) / 256, nz) : dim3((K + 255) / 256, (N + 127) / 128, nz);
    if (v2 && stages == 3)
      hipLaunchKernelGGL((gemm_tn_glds_kernel<2, 3>), grid, dim3(256), lds, (hipStream_t)stream, dY, ldy, X, ldx, out_partials, db_partials, M, N, K, mchunk, remap);
    else if (v2)
      hipLaunchKernelGGL((gemm_tn_glds_kernel<2, 5>), grid, dim3(256), lds, (hipStream_t)stream, dY, ldy, X, ldx, out_partials, db_partials, M, N, K, mchunk, remap);
    else if (stages == 3)
      hipLaunchKernelGGL((gemm_tn_glds_kernel<1, 3>), grid, dim3(256), lds, (hipStream_t)stream, dY, ldy, X, ldx, out_partials, db_partials, M, N, K, mchunk, remap);
    else
      hipLaunchKernelGGL((gemm_tn_glds_kernel<1, 5>), grid, dim3(256), lds, (hipStream_t)stream, dY, ldy, X, ldx, out_partials, db_partials, M, N, K, mchunk, remap);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(gemm_tn_x3_kernel, dim3((K + 127) / 128, (N + 127) / 128, nz), dim3(256), 0, (hipStream_t)stream, dY,
                     ldy, X, ldx, out_partials, db_partials, M, N, K, mchunk);
  return (int)hipGetLastError();
}

/* Weight gradient of a convolution over NHWC tokens as an implicit TN GEMM: dW[cout, (ky, kx, cin)] = sum over the output tokens
 * dY[t, cout] . X[input pixel of t under tap (ky, kx), cin]; ksize 1 or 3 (padding ksize / 2), stride 1 or 2 (output map ceil(Hin / 2) x
 * ceil(Win / 2)), Cin % 4 == 0, Cout >= 64.  Partials [splits, Cout, ksize^2 * Cin]; finish with combo_splitk_reduce_nchw_f32. */
int combo_conv_wgrad_x3_f32(const float* dY, long long ldy, const float* X, long long ldx, float* out_partials, int B, int Hin, int Win,
                            int Cin, int Cout, int ksize, int stride, int splits, combo_stream_t stream) {
  if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || Hin <= 0 || Win <= 0) return COMBO_EINVAL;
  const int H = (Hin + stride - 1) / stride, W = (Win + stride - 1) / stride;
  const long long M = (long long)B * H * W, Min = (long long)B * Hin * Win;
  const int K = ksize * ksize * Cin;
  if (!dY || !X || !out_partials || B <= 0 || (ksize == 3 && (Hin < 2 || Win < 2)) || M * (H > W ? H : W) >= (1LL << 32) || Min > 0x7fffffffLL / 4 ||
      Cin <= 0 || Cin % 4 != 0 || Cout < 64 || Cout % 4 != 0 || splits <= 0 || ldy % 4 != 0 || ldx % 4 != 0 ||
      ((uintptr_t)dY & 15) || ((uintptr_t)X & 15))
    return COMBO_EINVAL;
  int mchunk = (int)((M + splits - 1) / splits);
  mchunk = (mchunk + 15) / 16 * 16;
  if ((M + mchunk - 1) / mchunk != splits) return COMBO_EINVAL;
  constexpr int lds = 3 * (kTS * 256 * 4 + kTS * 128 * 4);
  static ComboDevFlag attr;
  if (!attr.is_set()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr.mark();
  }
  const int remap = 1;  // XCD-contiguous tile order
  // timing slot (bench.py `other_kernels`): useful flops 2 M Cout K; algorithmic bytes = dY and X read once + the partials written
  unsigned long long* ts = combo_timing_next_slot(COMBO_TS_CONV_WGRAD, 2.0 * M * Cout * K,
                                                  4.0 * ((double)M * Cout + (double)Min * Cin + (double)splits * Cout * K));
  TnConvGeom cg{H, W, Cin, (unsigned)(0xffffffffu / (unsigned)W + 1u), (unsigned)(0xffffffffu / (unsigned)H + 1u), Hin, Win, stride, ksize};
  if (tn_variant(Cout, K) == 2) {  // the less padded tile shape
    const dim3 grid((K + 127) / 128, (Cout + 255) / 256, splits);
    hipLaunchKernelGGL(conv3x3_wgrad_kernel<2>, grid, dim3(256), lds, (hipStream_t)stream, dY, ldy, X, ldx, out_partials, (int)M, Cout, K,
                       mchunk, remap, cg, ts);
  } else {
    const dim3 grid((K + 255) / 256, (Cout + 127) / 128, splits);
    hipLaunchKernelGGL(conv3x3_wgrad_kernel<1>, grid, dim3(256), lds, (hipStream_t)stream, dY, ldy, X, ldx, out_partials, (int)M, Cout, K,
                       mchunk, remap, cg, ts);
  }
  return (int)hipGetLastError();
}

int combo_conv3x3_wgrad_x3_f32(const float* dY, long long ldy, const float* X, long long ldx, float* out_partials, int B,
                               int H, int W, int Cin, int Cout, int splits, combo_stream_t stream) {
  return combo_conv_wgrad_x3_f32(dY, ldy, X, ldx, out_partials, B, H, W, Cin, Cout, 3, 1, splits, stream);
}

int combo_gemm_tn_x3_grouped_f32(const combo_gemm_tn_problem* problems, int count, combo_stream_t stream) {
  if (!problems || count <= 0) return COMBO_EINVAL;
  constexpr int lds = 3 * (kTS * 256 * 4 + kTS * 128 * 4);
  static ComboDevFlag attr;
  if (!attr.is_set()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_grouped_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr.mark();
  }
  const int remap = 1;  // XCD-contiguous tile order
  for (int base = 0; base < count; base += kMaxGroup) {
    TnGroupArgs a;
    a.count = count - base < kMaxGroup ? count - base : kMaxGroup;
    int blocks = 0;
    for (int i = 0; i < a.count; ++i) {
      const combo_gemm_tn_problem& pr = problems[base + i];
      if (!glds_ok(pr.dY, pr.ldy, pr.X, pr.ldx, pr.M, pr.N, pr.K) || !pr.partials || pr.splits <= 0) return COMBO_EINVAL;
      int mchunk = (pr.M + pr.splits - 1) / pr.splits;
      mchunk = (mchunk + 15) / 16 * 16;
      if ((pr.M + mchunk - 1) / mchunk != pr.splits) return COMBO_EINVAL;
      const long long tiles = tn_variant(pr.N, pr.K) == 2 ? (long long)((pr.N + 255) / 256) * ((pr.K + 127) / 128)
                                                         : (long long)((pr.N + 127) / 128) * ((pr.K + 255) / 256);
      a.block_start[i] = blocks;
      a.block_count[i] = (int)(tiles * pr.splits);
      a.p[i] = pr;
      blocks += a.block_count[i];
      if (remap) blocks = (blocks + 7) & ~7;
    }
    a.remap = remap;
    a.block_start[a.count] = blocks;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < a.count; ++i) {
      flops += 2.0 * a.p[i].M * a.p[i].N * a.p[i].K;
      bytes += 4.0 * ((double)a.p[i].M * (a.p[i].N + a.p[i].K) + (double)a.p[i].N * a.p[i].K);  // dY, X once; dW once
    }
    a.ts = combo_timing_next_slot(COMBO_TS_GEMM_TN, flops, bytes);
    hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, a);
  }
  return (int)hipGetLastError();
}

int combo_splitk_reduce_grouped_f32(const combo_reduce_problem* problems, int count, combo_stream_t stream) {
  if (!problems || count <= 0) return COMBO_EINVAL;
  for (int base = 0; base < count; base += kMaxGroup) {
    ReduceGroupArgs a;
    a.count = count - base < kMaxGroup ? count - base : kMaxGroup;
    long long threads = 0;
    for (int i = 0; i < a.count; ++i) {
      const combo_reduce_problem& pr = problems[base + i];
      if (!pr.partials || !pr.out || pr.splits <= 0 || pr.n <= 0 || (pr.n & 3) || ((uintptr_t)pr.partials & 15) ||
          ((uintptr_t)pr.out & 15) || (pr.nb > 0 && (!pr.db_partials || !pr.db)))
        return COMBO_EINVAL;
      a.thread_start[i] = threads;
      a.p[i] = pr;
      threads += (pr.n >> 2) + (pr.nb > 0 ? pr.nb : 0);
    }
    a.thread_start[a.count] = threads;
    hipLaunchKernelGGL(splitk_reduce_grouped_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
  }
  return (int)hipGetLastError();
}

}  // extern "C"
