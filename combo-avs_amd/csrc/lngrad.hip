// Deferred LayerNorm parameter gradients, grouped: dgamma[c] = sum_t dy[t,c] * (x[t,c] - mean[t]) * rstd[t],
// dbeta[c] = sum_t dy[t,c] for MANY LayerNorms in one launch.  ATen computes them per layer with two kernels
// (cuComputePartGradGammaBeta + cuComputeGradGammaBeta: ~100 launches / ~1 ms per step for the head); nothing on the
// backward critical path reads them, so ops/layernorm.py queues (dy, x, mean, rstd) and this kernel runs once at the end
// of the backward pass.  One workgroup = one (problem, token slice); threads walk the channels (coalesced), partial
// [slice][2][C] rows are finished by the grouped split-K reduce (gemm_tn.hip).
#include "combo_common.h"

namespace {

constexpr int kMaxLnGroup = 48;
struct LnGroupArgs {
  int count;
  int block_start[kMaxLnGroup + 1];
  combo_ln_grad_problem p[kMaxLnGroup];
};

// Round 5: (a) eight tokens in flight per thread - the serial walk had ONE pair of loads outstanding per thread and ran at
// 1.4 TB/s (1.8 ms of a `pvt_ms3_t10` step, 8 launches); (b) narrow rows use token sub-lanes: C = 64 / 128 kept 64 / 128 of the
// 256 threads busy - now thread (tl, c) takes the tokens t0 + tl, t0 + tl + ntl, ... (ntl = 256 / C) and the sub-lanes are summed
// through LDS in a fixed order.
__global__ void __launch_bounds__(256)
ln_param_grad_grouped_kernel(const LnGroupArgs args) {
  __shared__ float red[2][256];
  const int b = blockIdx.x;
  int pi = 0;
  for (int i = 1; i < args.count; ++i)
    if (b >= args.block_start[i]) pi = i;
  const combo_ln_grad_problem& pr = args.p[pi];
  const int slice = b - args.block_start[pi];
  const long long t0 = (long long)slice * pr.tokens_per_slice, t1 = min(pr.tokens, t0 + pr.tokens_per_slice);
  const int C = pr.C;
  const float* __restrict__ dy = pr.dy;
  const float* __restrict__ x = pr.x;
  const float* __restrict__ mean = pr.mean;
  const float* __restrict__ rstd = pr.rstd;
  const int ntl = (C <= 128 && 256 % C == 0) ? 256 / C : 1;  // token sub-lanes (uniform per workgroup)
  const int tl = ntl > 1 ? (int)threadIdx.x / C : 0;
  const int c0 = ntl > 1 ? (int)threadIdx.x % C : (int)threadIdx.x;
  float* o = pr.partials + (long long)slice * 2 * C;
  for (int c = c0; c < C; c += 256) {  // (ntl > 1: exactly one pass, every thread takes part in the barrier below)
    float dg = 0.f, db = 0.f;
    long long t = t0 + tl;
    for (; t + 7LL * ntl < t1; t += 8LL * ntl) {
      float g[8], xv[8], mu[8], rs[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long tt = t + (long long)u * ntl;
        g[u] = dy[tt * C + c];
        xv[u] = x[tt * C + c];
        mu[u] = mean[tt];
        rs[u] = rstd[tt];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        dg += g[u] * (xv[u] - mu[u]) * rs[u];
        db += g[u];
      }
    }
    for (; t < t1; t += ntl) {
      const float g = dy[t * C + c];
      dg += g * (x[t * C + c] - mean[t]) * rstd[t];
      db += g;
    }
    if (ntl > 1) {
      red[0][threadIdx.x] = dg;
      red[1][threadIdx.x] = db;
      __syncthreads();
      if (tl == 0) {
        for (int k = 1; k < ntl; ++k) { dg += red[0][k * C + c]; db += red[1][k * C + c]; }
        o[c] = dg;
        o[C + c] = db;
      }
    } else {
      o[c] = dg;
      o[C + c] = db;
    }
  }
}

}  // namespace

extern "C" int combo_ln_param_grad_grouped_f32(const combo_ln_grad_problem* problems, int count, combo_stream_t stream) {
  if (!problems || count <= 0) return COMBO_EINVAL;
  for (int base = 0; base < count; base += kMaxLnGroup) {
    LnGroupArgs a;
    a.count = count - base < kMaxLnGroup ? count - base : kMaxLnGroup;
    int blocks = 0;
    for (int i = 0; i < a.count; ++i) {
      const combo_ln_grad_problem& pr = problems[base + i];
      if (!pr.dy || !pr.x || !pr.mean || !pr.rstd || !pr.partials || pr.tokens <= 0 || pr.C <= 0 || pr.tokens_per_slice <= 0)
        return COMBO_EINVAL;
      a.block_start[i] = blocks;
      a.p[i] = pr;
      blocks += (int)((pr.tokens + pr.tokens_per_slice - 1) / pr.tokens_per_slice);
    }
    a.block_start[a.count] = blocks;
    hipLaunchKernelGGL(ln_param_grad_grouped_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  }
  return (int)hipGetLastError();
}
