// Linear sum assignment for the Hungarian matcher on the device, exact, for few targets per frame
// (reference: scipy.optimize.linear_sum_assignment at models/modeling/matcher.py:134, one host call per frame per
// decoder output after a .cpu() sync).  AVS frames carry G <= a handful of ground-truth instances against Q = 100
// queries.  In an optimal assignment the row chosen for a column is always among that column's G cheapest rows
// (otherwise it could be swapped for one of them that is unused), so the optimum is found by enumerating G^G
// candidate tuples: 4 for S4/MS3 (G = 2), 46 656 at the supported maximum G = 6.  One wave per problem; no host
// round trip, so the training step has no device->host synchronisation left.  Ties are broken towards the smaller
// candidate index (scipy's choice among equal-cost optima is unspecified as well).
// Non-finite costs (a diverged step: NaN / Inf logits) are read as the largest finite cost, so the kernel ALWAYS returns valid,
// distinct query indices - scipy raises ValueError there (matcher.py:133); here the NaN simply reaches the loss, and the
// downstream gathers (criterion mask_index) can never be driven out of bounds (round 1: a NaN cost left the arg-min at its
// 0x7fffffff sentinel, which the mask-loss kernels then used as a row index - a hardware exception, not an error message).
// The condition is SURFACED, not hidden: `status` (nullable device word) gets bit 0 when a non-finite cost was read and bit 1
// when a frame's target count exceeds what this kernel solves (it then solves the first min(Gpad, 6) targets); the host side
// (HungarianMatcher.check_status) turns either into the ValueError scipy would have raised, without a per-step sync.
#include <math.h>

#include "combo_common.h"

namespace {

constexpr int GMAX = 6;

__global__ void __launch_bounds__(64)
lsap_small_kernel(const float* __restrict__ cost, const int* __restrict__ gcount, int N, int Q, int Gpad,
                  long long* __restrict__ row_for_col, int* __restrict__ status) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const int G = min(max(gcount[n], 0), min(Gpad, GMAX));
  bool bad = false;
  if (status && lane == 0 && gcount[n] > min(Gpad, GMAX)) atomicOr(status, COMBO_LSAP_TOO_MANY_TARGETS);
  const float* C = cost + (long long)n * Q * Gpad;
  __shared__ int cand[GMAX][GMAX];
  __shared__ float candc[GMAX][GMAX];
  // ---- the G cheapest rows of every column (G rounds of a wave arg-min with exclusion) ----
  for (int g = 0; g < G; ++g) {
    for (int r = 0; r < G; ++r) {
      float best = INFINITY;
      int bi = 0x7fffffff;
      for (int q = lane; q < Q; q += 64) {
        bool used = false;
        for (int p = 0; p < r; ++p) used |= (cand[g][p] == q);
        float c = C[q * Gpad + g];
        const bool fin = (c == c && fabsf(c) < 3.0e38f);
        bad |= !fin;
        c = fin ? c : 3.0e38f;  // NaN / Inf -> the largest finite cost
        if (!used && (c < best || (c == best && q < bi))) { best = c; bi = q; }
      }
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) {
        const float ob = __shfl_xor(best, s);
        const int oi = __shfl_xor(bi, s);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (lane == 0) { cand[g][r] = bi; candc[g][r] = best; }
      __syncthreads();
    }
  }
  if (status && __any(bad) && lane == 0) atomicOr(status, COMBO_LSAP_NONFINITE_COST);
  // ---- enumerate the G^G tuples ----
  int total = 1;
  for (int g = 0; g < G; ++g) total *= G;
  float best = INFINITY;
  int bt = 0x7fffffff;
  for (int t = lane; t < total; t += 64) {
    int idx = t, rows[GMAX];
    float c = 0.f;
    bool ok = true;
    for (int g = 0; g < G; ++g) {
      const int r = idx % G;
      idx /= G;
      rows[g] = cand[g][r];
      c += candc[g][r];
      for (int p = 0; p < g; ++p) ok &= (rows[p] != rows[g]);
    }
    if (ok && (c < best || (c == best && t < bt))) { best = c; bt = t; }
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    const float ob = __shfl_xor(best, s);
    const int oi = __shfl_xor(bt, s);
    if (ob < best || (ob == best && oi < bt)) { best = ob; bt = oi; }
  }
  if (lane < Gpad) {
    long long out = -1;
    if (lane < G) {
      int idx = bt;
      for (int g = 0; g < lane; ++g) idx /= G;
      out = cand[lane][idx % G];
      out = out < 0 ? 0 : (out >= Q ? Q - 1 : out);  // (unreachable with the sanitised costs; never an out-of-range row)
    }
    row_for_col[(long long)n * Gpad + lane] = out;
  }
}

}  // namespace

extern "C" int combo_lsap_small_f32(const float* cost, const int* gcount, int N, int Q, int Gpad, long long* row_for_col,
                                    int* status, combo_stream_t stream) {
  if (!cost || !gcount || !row_for_col || N <= 0 || Q <= 0 || Gpad <= 0 || Gpad > GMAX || Q < Gpad) return COMBO_EINVAL;
  hipLaunchKernelGGL(lsap_small_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, cost, gcount, N, Q, Gpad, row_for_col, status);
  return (int)hipGetLastError();
}
