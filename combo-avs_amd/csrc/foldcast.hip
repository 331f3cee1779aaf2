// Multi-tensor "scale per output channel + cast": dst_i = cast(src_i * scale_i[channel]) for MANY tensors in one launch.
// Used for (a) folding FrozenBN into every convolution weight of a ResNet and casting it to the compute dtype, and the way
// back for the gradients (backbone._FoldAll), (b) the per-step bf16 copies of all PVTv2 parameters (backbone_pvt._CastAll).
// torch._foreach_mul / _foreach_copy_ fall back to ONE kernel per tensor when shapes broadcast or dtypes differ (measured:
// 2 x 106 launches forward and backward for the two ResNet-50s); here the tensor table travels in the kernel arguments.
#include "combo_common.h"

namespace {

constexpr int kMaxFold = 56;
constexpr int kChunk = 2048;  // elements per workgroup
struct FoldArgs {
  int count;
  int block_start[kMaxFold + 1];
  combo_fold_problem p[kMaxFold];
};

__device__ __forceinline__ float ld(const void* p, long long i, int bf16) {
  if (bf16) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(p)[i] << 16);
  return reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void st(void* p, long long i, float v, int bf16) {
  if (bf16) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) u = (u >> 16) | 0x40;
    else { u += 0x7fffu + ((u >> 16) & 1u); u >>= 16; }
    reinterpret_cast<unsigned short*>(p)[i] = (unsigned short)u;
  } else {
    reinterpret_cast<float*>(p)[i] = v;
  }
}

__global__ void __launch_bounds__(256)
fold_cast_grouped_kernel(const FoldArgs args) {
  const int b = blockIdx.x;
  int pi = 0;
  for (int i = 1; i < args.count; ++i)
    if (b >= args.block_start[i]) pi = i;
  const combo_fold_problem& pr = args.p[pi];
  const long long e0 = (long long)(b - args.block_start[pi]) * kChunk;
  const long long e1 = min(pr.numel, e0 + kChunk);
  for (long long e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
    float v = ld(pr.src, e, pr.src_bf16);
    if (pr.scale) v *= pr.scale[e / pr.inner];
    st(pr.dst, e, v, pr.dst_bf16);
  }
}

}  // namespace

extern "C" int combo_fold_cast_grouped(const combo_fold_problem* problems, int count, combo_stream_t stream) {
  if (!problems || count <= 0) return COMBO_EINVAL;
  for (int base = 0; base < count; base += kMaxFold) {
    FoldArgs a;
    a.count = count - base < kMaxFold ? count - base : kMaxFold;
    int blocks = 0;
    for (int i = 0; i < a.count; ++i) {
      const combo_fold_problem& pr = problems[base + i];
      if (!pr.src || !pr.dst || pr.numel <= 0 || pr.inner <= 0) return COMBO_EINVAL;
      a.block_start[i] = blocks;
      a.p[i] = pr;
      blocks += (int)((pr.numel + kChunk - 1) / kChunk);
    }
    a.block_start[a.count] = blocks;
    hipLaunchKernelGGL(fold_cast_grouped_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  }
  return (int)hipGetLastError();
}
