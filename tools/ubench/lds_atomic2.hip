// micro-benchmark: LDS integer atomic throughput on gfx950 (u32 / u64, with and without return)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T, int MODE, bool RTN>
__global__ void __launch_bounds__(1024) k(T* out, const int* idx, int iters, long long* cyc) {
  extern __shared__ char smem[];
  T* lds = (T*)smem;
  const int N = 65536 / sizeof(T);
  const int tid = threadIdx.x;
  for (int i = tid; i < N; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  int a[16];
  for (int j = 0; j < 16; ++j) {
    if (MODE == 0) a[j] = (tid + j * 1024) & (N - 1);
    else if (MODE == 2) a[j] = idx[(tid * 16 + j) & 65535] & (N - 1);
    else if (MODE == 3) a[j] = idx[(tid * 16 + j) & 65535] & 255;   // hot: 256 addresses
  }
  T acc = 0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (RTN) acc += __hip_atomic_fetch_add(&lds[a[j]], (T)(tid + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(&lds[a[j]], (T)(tid + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + tid] = lds[tid] + acc;
}

int main() {
  unsigned long long* out; int* idx; long long* cyc;
  hipMalloc(&out, 1024 * 256 * 8); hipMalloc(&idx, 65536 * 4); hipMalloc(&cyc, 256 * 8);
  std::vector<int> h(65536); unsigned s = 12345;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (int)(s >> 8); }
  hipMemcpy(idx, h.data(), 65536 * 4, hipMemcpyHostToDevice);
  const int iters = 200;
  auto run = [&](auto kern, auto* o, const char* name, int threads) {
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 65536, 0, o, idx, iters, cyc);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-40s threads=%4d: %8.3f cycles/wave-instr  %6.3f cycles/lane-op\n", name, threads,
           (double)c / (iters * 16.0 * (threads / 64)), (double)c / ((double)iters * 16 * threads));
  };
  for (int th : {64, 1024}) {
    run(k<unsigned, 0, false>, (unsigned*)out, "ds_add_u32 distinct banks", th);
    run(k<unsigned, 2, false>, (unsigned*)out, "ds_add_u32 random 64KB", th);
    run(k<unsigned, 3, false>, (unsigned*)out, "ds_add_u32 random hot(256 addr)", th);
    run(k<unsigned, 2, true>, (unsigned*)out, "ds_add_rtn_u32 random 64KB", th);
    run(k<unsigned long long, 0, false>, out, "ds_add_u64 distinct banks", th);
    run(k<unsigned long long, 2, false>, out, "ds_add_u64 random 64KB", th);
    run(k<unsigned long long, 3, false>, out, "ds_add_u64 random hot(256 addr)", th);
    run(k<float, 2, false>, (float*)out, "ds_add_f32 random 64KB", th);
    run(k<double, 2, false>, (double*)out, "ds_add_f64 random 64KB", th);
  }
  return 0;
}
