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

__global__ void __launch_bounds__(256)
ln_param_grad_grouped_kernel(const LnGroupArgs args) {
  const int b = blockIdx.x;
  int pi = 0;
  for (int i = 1; i < args.count; ++i)
    if (b >= args.block_start[i]) pi = i;
  const combo_ln_grad_problem& pr = args.p[pi];
  const int slice = b - args.block_start[pi];
  const long long t0 = (long long)slice * pr.tokens_per_slice, t1 = min(pr.tokens, t0 + pr.tokens_per_slice);
  const int C = pr.C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float dg = 0.f, db = 0.f;
    for (long long t = t0; t < t1; ++t) {
      const float g = pr.dy[t * C + c];
      dg += g * (pr.x[t * C + c] - pr.mean[t]) * pr.rstd[t];
      db += g;
    }
    float* o = pr.partials + (long long)slice * 2 * C;
    o[c] = dg;
    o[C + c] = db;
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
