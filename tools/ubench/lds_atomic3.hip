// micro-benchmark: ds_add_u32 vs ds_add_u64 throughput per CU on gfx950 (one 768-thread workgroup = the grad_value
// kernel's shape), with the address patterns of csrc/msda.hip's backward: (a) v1: 16 random rows x 4 lanes x 1 int,
// (b) v2: 2 random rows x 32 consecutive ints (conflict-free), (c) 4 random rows x 16 consecutive int64 (two channels
// packed per 64-bit add), (d) distinct consecutive addresses.  Prints LDS cycles per wave instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T, int MODE>
__global__ void __launch_bounds__(768) k(T* out, const int* idx, int iters, long long* cyc) {
  extern __shared__ char smem[];
  T* lds = reinterpret_cast<T*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int n = 131072 / sizeof(T);
  for (int i = tid; i < n; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  int a[16];
  for (int j = 0; j < 16; ++j) {
    const int rnd = idx[(tid * 16 + j) & 65535], wrnd = idx[((tid >> 6) * 16 + j + 7) & 65535];
    if (MODE == 0) a[j] = (tid + j * 768) % n;                                        // distinct consecutive
    else if (MODE == 1) a[j] = ((idx[((tid >> 2) * 16 + j) & 65535] & 1023) * 16 + (lane & 3) * 4 + (j & 3)) % n;  // v1
    else if (MODE == 2) a[j] = (((wrnd >> (lane >> 5) * 10) & 1023) * 32 + (lane & 31)) % n;   // 2 rows x 32 ints
    else if (MODE == 3) a[j] = (((wrnd >> (lane >> 4) * 5) & 1023) * 16 + (lane & 15)) % n;    // 4 rows x 16 elements
    (void)rnd;
  }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) __hip_atomic_fetch_add(&lds[a[j]], (T)(it + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + tid] = lds[tid];
}

int main() {
  void* out; int* idx; long long* cyc;
  hipMalloc(&out, 768 * 256 * 8); hipMalloc(&idx, 65536 * 4); hipMalloc(&cyc, 256 * 8);
  std::vector<int> h(65536); unsigned s = 12345;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (int)(s >> 8); }
  hipMemcpy(idx, h.data(), 65536 * 4, hipMemcpyHostToDevice);
  const int iters = 200, threads = 768;
  auto run = [&](auto kern, auto* o, const char* name) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 131072, 0, o, idx, iters, cyc);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-52s %8.2f cycles per wave instruction (12 waves on the CU)\n", name, (double)c / (iters * 16.0 * (threads / 64)));
  };
  run(k<unsigned, 0>, (unsigned*)out, "ds_add_u32 consecutive");
  run(k<unsigned, 1>, (unsigned*)out, "ds_add_u32 v1: 16 rows x 4 lanes");
  run(k<unsigned, 2>, (unsigned*)out, "ds_add_u32 v2: 2 rows x 32 ints");
  run(k<unsigned long long, 0>, (unsigned long long*)out, "ds_add_u64 consecutive");
  run(k<unsigned long long, 3>, (unsigned long long*)out, "ds_add_u64 4 rows x 16 int64");
  run(k<unsigned long long, 2>, (unsigned long long*)out, "ds_add_u64 2 rows x 32 int64");
  return 0;
}
