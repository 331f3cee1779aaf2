// Channel sums  out[c] = sum_{a, l} x[a, c, l]  of a contiguous [A, C, L] tensor (bf16 or fp32): the bias gradients of the
// host-PyTorch backbones' linears / convolutions (L = 1: column sums of dY[M, C]; L = H*W: NCHW convolution outputs) and the
// level-embedding gradients of the head.
//
// Why not ATen's sum: when one output is split over several workgroups ATen (Reduce.cuh) zeroes its semaphores with a
// hipMemsetAsync per launch; the HIP runtime's AQL packet capture replays such memset nodes of a hipGraph wrongly on this
// stack (ROCm 7.2, torch 2.10: tools/graph_reduce_repro.py - 99 of 100 replays return stale memory), so the captured
// training step must hold none.  Here a first launch writes partial[slice][c] and a second, tiny one adds the slices in a
// fixed order: no memset, no atomics, deterministic.  (v1 did it in one launch with "last workgroup to arrive adds the
// partials": the agent-scope release/acquire fence every workgroup needs for that costs ~17 ns and serialises chip-wide -
// 120-140 us per launch at 2 048 workgroups, whatever the size of the input; the kernel boundary does the same job once.)
//
// HBM-bound: x is read once (A*C*L*elem bytes), everything else is C-sized.
#include "combo_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTargetGroups = 2048;  // workgroups to aim for: 8 per CU
constexpr int kMaxSlices = 256;
constexpr int kFinishThreads = 1024;

__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned short to_bf16(float v) {
  unsigned u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

struct ColArgs {
  const void* x;
  void* out;
  float* partial;  // [slices][C]
  long long A, L;
  int C, slices, in_bf16, out_bf16;
  int tx_log2;     // rows mode: lanes along the columns = 1 << tx_log2 (16, 32 or 64), each owning 4 columns
};

__device__ __forceinline__ void store_out(const ColArgs& a, int c, float v) {
  if (a.out_bf16) reinterpret_cast<unsigned short*>(a.out)[c] = to_bf16(v);
  else reinterpret_cast<float*>(a.out)[c] = v;
}

// L == 1: x[M, C], C % 4 == 0.  Workgroup = tx lanes along the columns (4 columns each) x ty = 256 / tx rows in flight;
// grid = (column blocks, slices of the rows).
__device__ __forceinline__ void colsum_rows_body(const ColArgs& a, int bx, int by, float4* red) {
  const int tx = 1 << a.tx_log2, ty = kThreads >> a.tx_log2;
  const int lx = threadIdx.x & (tx - 1), ly = threadIdx.x >> a.tx_log2;
  const int col = (bx * tx + lx) * 4;
  const long long M = a.A;
  const long long rows_per = (M + a.slices - 1) / a.slices;
  const long long r0 = rows_per * by, r1 = min(M, r0 + rows_per);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < a.C) {
    const long long pitch = a.C / 4;  // in 4-column groups
    long long r = r0 + ly;
    if (a.in_bf16) {
      const uint2* p = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.x) + col);
      for (; r + 3ll * ty < r1; r += 4ll * ty) {  // four independent loads in flight
        const uint2 v0 = p[r * pitch], v1 = p[(r + ty) * pitch], v2 = p[(r + 2 * ty) * pitch], v3 = p[(r + 3 * ty) * pitch];
        acc.x += (bf16_lo(v0.x) + bf16_lo(v1.x)) + (bf16_lo(v2.x) + bf16_lo(v3.x));
        acc.y += (bf16_hi(v0.x) + bf16_hi(v1.x)) + (bf16_hi(v2.x) + bf16_hi(v3.x));
        acc.z += (bf16_lo(v0.y) + bf16_lo(v1.y)) + (bf16_lo(v2.y) + bf16_lo(v3.y));
        acc.w += (bf16_hi(v0.y) + bf16_hi(v1.y)) + (bf16_hi(v2.y) + bf16_hi(v3.y));
      }
      for (; r < r1; r += ty) {
        const uint2 v = p[r * pitch];
        acc.x += bf16_lo(v.x); acc.y += bf16_hi(v.x); acc.z += bf16_lo(v.y); acc.w += bf16_hi(v.y);
      }
    } else {
      const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.x) + col);
      for (; r + 3ll * ty < r1; r += 4ll * ty) {
        const float4 v0 = p[r * pitch], v1 = p[(r + ty) * pitch], v2 = p[(r + 2 * ty) * pitch], v3 = p[(r + 3 * ty) * pitch];
        acc.x += (v0.x + v1.x) + (v2.x + v3.x); acc.y += (v0.y + v1.y) + (v2.y + v3.y);
        acc.z += (v0.z + v1.z) + (v2.z + v3.z); acc.w += (v0.w + v1.w) + (v2.w + v3.w);
      }
      for (; r < r1; r += ty) {
        const float4 v = p[r * pitch];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = ty >> 1; s > 0; s >>= 1) {  // over the rows in flight, fixed tree
    if (ly < s) {
      const float4 o = red[threadIdx.x + s * tx];
      acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
      red[threadIdx.x] = acc;
    }
    __syncthreads();
  }
  if (ly != 0 || col >= a.C) return;
  if (a.slices == 1) { store_out(a, col, acc.x); store_out(a, col + 1, acc.y); store_out(a, col + 2, acc.z); store_out(a, col + 3, acc.w); }
  else *reinterpret_cast<float4*>(a.partial + (long long)by * a.C + col) = acc;
}

__global__ void __launch_bounds__(kThreads)
colsum_rows_kernel(const ColArgs a) {
  __shared__ float4 red[kThreads];
  colsum_rows_body(a, blockIdx.x, blockIdx.y, red);
}

// Many small column sums in ONE launch (+ one finish launch): the ~630 bias gradients of two PVTv2-B5 backbones cost 5 + 5 us
// each as separate launches (6.4 ms of a 140 ms step) for 1 - 20 MB of input.  The problem table travels in the kernel
// arguments; a workgroup finds its problem by its block index.
constexpr int kMaxGroup = 40;
struct GroupArgs {
  int count;
  int block_start[kMaxGroup + 1];   // rows kernel: first workgroup of problem i
  int finish_start[kMaxGroup + 1];  // finish kernel: likewise
  int blocks_x[kMaxGroup];
  ColArgs p[kMaxGroup];
};

__device__ __forceinline__ int find_problem(const int* start, int count, int b) {
  int pi = 0;
  for (int i = 1; i < count; ++i)
    if (b >= start[i]) pi = i;
  return pi;
}

__global__ void __launch_bounds__(kThreads)
colsum_rows_grouped_kernel(const GroupArgs g) {
  __shared__ float4 red[kThreads];
  const int pi = find_problem(g.block_start, g.count, blockIdx.x);
  const int lb = blockIdx.x - g.block_start[pi];
  colsum_rows_body(g.p[pi], lb % g.blocks_x[pi], lb / g.blocks_x[pi], red);
}

// L > 1: x[A, C, L].  Workgroup = (channel c, slice of the A planes); waves take planes in turn, lanes run along L.
template <int V>
__global__ void __launch_bounds__(kThreads)
colsum_planes_kernel(const ColArgs a) {
  __shared__ float red[kThreads / COMBO_WAVE];
  const int c = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long per = (a.A + a.slices - 1) / a.slices;
  const long long a0 = per * blockIdx.y, a1 = min(a.A, a0 + per);
  float acc = 0.f;
  for (long long pl = a0 + wave; pl < a1; pl += kThreads / COMBO_WAVE) {
    const long long base = (pl * a.C + c) * a.L;
    if (a.in_bf16) {
      const unsigned short* p = reinterpret_cast<const unsigned short*>(a.x) + base;
      if (V == 4) {
        for (long long l = lane * 4ll; l < a.L; l += 256) {
          const uint2 v = *reinterpret_cast<const uint2*>(p + l);
          acc += (bf16_lo(v.x) + bf16_hi(v.x)) + (bf16_lo(v.y) + bf16_hi(v.y));
        }
      } else {
        for (long long l = lane; l < a.L; l += 64) acc += __uint_as_float((unsigned)p[l] << 16);
      }
    } else {
      const float* p = reinterpret_cast<const float*>(a.x) + base;
      if (V == 4) {
        for (long long l = lane * 4ll; l < a.L; l += 256) {
          const float4 v = *reinterpret_cast<const float4*>(p + l);
          acc += (v.x + v.y) + (v.z + v.w);
        }
      } else {
        for (long long l = lane; l < a.L; l += 64) acc += p[l];
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const float total = (red[0] + red[1]) + (red[2] + red[3]);
  if (a.slices == 1) store_out(a, c, total);
  else a.partial[(long long)blockIdx.y * a.C + c] = total;
}

// out[c] = sum_s partial[s][c]: 64 columns per workgroup, 16 row groups; row group ly adds slices ly, ly + 16, ... in ascending
// order, then a fixed tree over the row groups.
__device__ __forceinline__ void colsum_finish_body(const ColArgs& a, int bx, float* red) {
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  const int c = bx * 64 + lx;
  float t = 0.f;
  if (c < a.C)
    for (int s = ly; s < a.slices; s += kFinishThreads / 64) t += a.partial[(long long)s * a.C + c];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int s = kFinishThreads / 128; s > 0; s >>= 1) {
    if (ly < s) { t += red[threadIdx.x + s * 64]; red[threadIdx.x] = t; }
    __syncthreads();
  }
  if (ly == 0 && c < a.C) store_out(a, c, t);
}

__global__ void __launch_bounds__(kFinishThreads)
colsum_finish_kernel(const ColArgs a) {
  __shared__ float red[kFinishThreads];
  colsum_finish_body(a, blockIdx.x, red);
}

__global__ void __launch_bounds__(kFinishThreads)
colsum_finish_grouped_kernel(const GroupArgs g) {
  __shared__ float red[kFinishThreads];
  const int pi = find_problem(g.finish_start, g.count, blockIdx.x);
  if (g.p[pi].slices > 1) colsum_finish_body(g.p[pi], blockIdx.x - g.finish_start[pi], red);
}

struct Plan { int blocks_x, slices, tx_log2; };

Plan plan(long long A, int C, long long L) {
  Plan p{};
  if (L == 1) {
    const int groups = C / 4;
    // lanes along the columns: the width (16 / 32 / 64) that wastes the fewest lanes, wider on ties
    int best = 64, best_waste = 1 << 30;
    for (int tx = 64; tx >= 16; tx >>= 1) {
      const int waste = (groups + tx - 1) / tx * tx - groups;
      if (waste < best_waste) { best = tx; best_waste = waste; }
    }
    p.tx_log2 = best == 64 ? 6 : best == 32 ? 5 : 4;
    p.blocks_x = (groups + best - 1) / best;
    const int ty = kThreads / best;
    long long s = kTargetGroups / p.blocks_x;
    const long long max_s = A / (8ll * ty);  // at least 8 rows per thread and slice
    if (s > max_s) s = max_s;
    p.slices = (int)(s < 1 ? 1 : s > kMaxSlices ? kMaxSlices : s);
  } else {
    p.blocks_x = C;
    long long s = kTargetGroups / (C > 0 ? C : 1);
    if (s > A) s = A;
    p.slices = (int)(s < 1 ? 1 : s > kMaxSlices ? kMaxSlices : s);
  }
  return p;
}

// inside a grouped launch a problem takes fewer slices (the launch as a whole fills the chip)
int grouped_slices(const Plan& p, long long A) {
  const int ty = kThreads >> p.tx_log2;
  long long s = 256 / p.blocks_x;
  const long long max_s = A / (8ll * ty);
  if (s > max_s) s = max_s;
  return (int)(s < 1 ? 1 : s > 64 ? 64 : s);
}

}  // namespace

extern "C" {

int combo_colsum_grouped_slices(long long rows, int C) {
  if (rows <= 0 || C <= 0 || C % 4 != 0) return COMBO_EINVAL;
  return grouped_slices(plan(rows, C, 1), rows);
}

int combo_colsum_grouped(const combo_colsum_problem* problems, int count, combo_stream_t stream) {
  if (!problems || count <= 0) return COMBO_EINVAL;
  for (int base = 0; base < count; base += kMaxGroup) {
    GroupArgs g;
    g.count = count - base < kMaxGroup ? count - base : kMaxGroup;
    int blocks = 0, fin = 0;
    bool any_finish = false;
    for (int i = 0; i < g.count; ++i) {
      const combo_colsum_problem& pr = problems[base + i];
      if (!pr.x || !pr.out || pr.rows <= 0 || pr.C <= 0 || pr.C % 4 != 0) return COMBO_EINVAL;
      const Plan p = plan(pr.rows, pr.C, 1);
      const int slices = grouped_slices(p, pr.rows);
      if (slices > 1 && !pr.partial) return COMBO_EINVAL;
      g.p[i] = ColArgs{pr.x, pr.out, pr.partial, pr.rows, 1, pr.C, slices, pr.in_bf16, pr.out_bf16, p.tx_log2};
      g.blocks_x[i] = p.blocks_x;
      g.block_start[i] = blocks;
      g.finish_start[i] = fin;
      blocks += p.blocks_x * slices;
      fin += (pr.C + 63) / 64;
      any_finish |= slices > 1;
    }
    g.block_start[g.count] = blocks;
    g.finish_start[g.count] = fin;
    hipLaunchKernelGGL(colsum_rows_grouped_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, g);
    if (any_finish) hipLaunchKernelGGL(colsum_finish_grouped_kernel, dim3(fin), dim3(kFinishThreads), 0, (hipStream_t)stream, g);
  }
  return (int)hipGetLastError();
}

int combo_colsum_slices(long long A, int C, long long L) {
  if (A <= 0 || C <= 0 || L <= 0 || (L == 1 && C % 4 != 0)) return COMBO_EINVAL;
  return plan(A, C, L).slices;
}

int combo_colsum(const void* x, long long A, int C, long long L, int in_bf16, void* out, int out_bf16, float* partial,
                 combo_stream_t stream) {
  if (!x || !out || A <= 0 || C <= 0 || L <= 0 || (L == 1 && C % 4 != 0)) return COMBO_EINVAL;
  const Plan p = plan(A, C, L);
  if (p.slices > 1 && !partial) return COMBO_EINVAL;
  ColArgs a{x, out, partial, A, L, C, p.slices, in_bf16, out_bf16, p.tx_log2};
  const dim3 grid(p.blocks_x, p.slices);
  if (L == 1) hipLaunchKernelGGL(colsum_rows_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, a);
  else if (L % 4 == 0) hipLaunchKernelGGL(colsum_planes_kernel<4>, grid, dim3(kThreads), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(colsum_planes_kernel<1>, grid, dim3(kThreads), 0, (hipStream_t)stream, a);
  if (p.slices > 1)
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 63) / 64), dim3(kFinishThreads), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // extern "C"
