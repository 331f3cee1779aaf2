// micro-benchmark: LDS float atomic-add throughput on gfx950 under different address patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, const int* idx, int iters, long long* cyc) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 16384; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  int a[16];
  for (int j = 0; j < 16; ++j) {
    if (MODE == 0) a[j] = (tid + j * 1024) & 16383;                 // distinct banks, distinct addresses
    else if (MODE == 1) a[j] = j;                                    // all lanes same address
    else if (MODE == 2) a[j] = idx[(tid * 16 + j) & 65535] & 16383;  // random in 64 KB
    else if (MODE == 3) a[j] = idx[(tid * 16 + j) & 65535] & 1023;   // random in 4 KB (hot rows)
    else if (MODE == 4) a[j] = ((tid >> 2) * 16 + (tid & 3) * 4 + (j & 3)) & 16383;  // v1 kernel pattern
    else if (MODE == 5) a[j] = (idx[(tid >> 2) & 65535] & 1023) * 16 + (tid & 3) * 4 + (j & 3);  // random rows, 4 lanes/row
  }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
      __hip_atomic_fetch_add(&lds[a[j]], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + tid] = lds[tid];
}

template <int MODE>
__global__ void __launch_bounds__(1024) kw(float* out, const int* idx, int iters, long long* cyc) {
  // same with plain ds_write_b32 for comparison
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  int a[16];
  for (int j = 0; j < 16; ++j) a[j] = (MODE == 0) ? ((tid + j * 1024) & 16383) : (idx[(tid * 16 + j) & 65535] & 16383);
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) ((volatile float*)lds)[a[j]] = (float)it;
  }
  __syncthreads();
  long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + tid] = lds[tid];
}

int main() {
  float* out; int* idx; long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&idx, 65536 * 4); hipMalloc(&cyc, 256 * 8);
  std::vector<int> h(65536); unsigned s = 12345;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (int)(s >> 8); }
  hipMemcpy(idx, h.data(), 65536 * 4, hipMemcpyHostToDevice);
  const int iters = 200;
  auto run = [&](auto kern, const char* name, int threads) {
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 65536, 0, out, idx, iters, cyc);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double ops = (double)iters * 16 * threads;  // lane-ops
    printf("%-44s threads=%4d: %8.3f cycles/wave-instr  %6.3f cycles/lane-op\n", name, threads,
           (double)c / (iters * 16.0 * (threads / 64)), (double)c / ops);
  };
  for (int th : {64, 256, 1024}) {
    run(k<0>, "ds_add_f32 distinct banks", th);
    run(k<1>, "ds_add_f32 same address", th);
    run(k<2>, "ds_add_f32 random 64KB", th);
    run(k<3>, "ds_add_f32 random 4KB", th);
    run(k<4>, "ds_add_f32 v1 pattern (8 banks)", th);
    run(k<5>, "ds_add_f32 random rows x 4 lanes", th);
    run(kw<0>, "ds_write_b32 distinct banks", th);
    run(kw<2>, "ds_write_b32 random 64KB", th);
  }
  return 0;
}
