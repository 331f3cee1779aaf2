// micro-benchmark (round 5): ds_add_u64 with the two scatter forms of csrc/msda_bwd.hip.
//   record form : a wave instruction = 4 samples x 16 channel pairs = 4 rows x 128 contiguous bytes
//   rotated form: a wave instruction = 64 samples, lane l adds channel pair (l + t) & 15 of ITS row (64 rows per instruction,
//                 every 16-lane group covers 16 distinct bank pairs)
// over `rows` accumulator rows (1024: a band of a large level; 196: the 14 x 14 level; 49: the 7 x 7 level - same-address collisions),
// and with the 4 samples of a query on ONE row (zero-initialised sampling offsets: the bench step's first iterations).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(512) k(unsigned long long* out, const int* idx, int iters, int rows, int same_query, long long* cyc) {
  extern __shared__ char smem[];
  unsigned long long* lds = reinterpret_cast<unsigned long long*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < rows * 16; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  int a[16];
  for (int j = 0; j < 16; ++j) {
    if (MODE == 0) {  // record form: instruction j serves samples 4 j' .. ; lane (u4 = lane >> 4, cp = lane & 15)
      const int sample = same_query ? (wave * 64 + j * 4) / 4 : wave * 64 + j * 4 + (lane >> 4);
      a[j] = (idx[(sample * 7 + j) & 65535] % rows) * 16 + (lane & 15);
    } else {          // rotated form: lane = sample, step j -> channel pair (lane + j) & 15
      const int sample = same_query ? (wave * 64 + lane) / 4 : wave * 64 + lane;
      a[j] = (idx[(sample * 7) & 65535] % rows) * 16 + ((lane + j) & 15);
    }
  }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) __hip_atomic_fetch_add(&lds[a[j]], (unsigned long long)(it + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + tid] = lds[tid % (rows * 16)];
}

int main() {
  void* out; int* idx; long long* cyc;
  hipMalloc(&out, 512 * 256 * 8); hipMalloc(&idx, 65536 * 4); hipMalloc(&cyc, 256 * 8);
  std::vector<int> h(65536); unsigned s = 12345;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (int)(s >> 8); }
  hipMemcpy(idx, h.data(), 65536 * 4, hipMemcpyHostToDevice);
  const int iters = 200, threads = 512;
  auto run = [&](auto kern, int rows, int same, const char* name) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 131072, 0, (unsigned long long*)out, idx, iters, rows, same, cyc);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s rows %4d %s %8.2f cycles per wave instruction (8 waves on the CU)\n", name, rows, same ? "4 samples of a query on one row" : "independent rows              ",
           (double)c / (iters * 16.0 * (threads / 64)));
  };
  for (int same = 0; same < 2; ++same)
    for (int rows : {1024, 196, 49}) {
      run(k<0>, rows, same, "ds_add_u64 record form (4 x 16)");
      run(k<1>, rows, same, "ds_add_u64 rotated form (64 rows)");
    }
  return 0;
}
