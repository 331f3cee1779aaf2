// Does v_mfma_f32_32x32x16_f16 on gfx950 keep fp16 SUBNORMAL inputs (the lo piece of an fp16 hi / lo split is subnormal for |x| < 2^-3)?
// and what do v_cvt_pkrtz_f16_f32 / v_cvt_pk_f16_f32 do with values below the fp16 normal range?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_f16_denorm.hip -o tools/ubench/mfma_f16_denorm && tools/ubench/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

__global__ void probe(const float* av, const float* bv, float* out, unsigned* cv) {
  const int lane = threadIdx.x;
  half8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = (_Float16)0.f; b[k] = (_Float16)0.f; }
  // row r = lane & 31 of A (second operand) x column n = lane & 31 of B; k-half g = lane >> 5: put the test value at k = 0 of g = 0
  if (lane < 32) {
    half2v h = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(av[lane], 0.f));
    a[0] = h[0];
    b[0] = (_Float16)bv[0];
    cv[lane] = (unsigned)__builtin_bit_cast(unsigned short, h[0]);
  }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c, 0, 0, 0);
  // D[n][m]: lane (m = lane & 31, g) holds n = 8q + 4g + e: n = 0 is c[0] of lanes 0..31
  if (lane < 32) out[lane] = c[0];
}

int main() {
  float ha[32], hb[1] = {1.0f}, ho[32];
  unsigned hc[32];
  for (int i = 0; i < 32; ++i) ha[i] = ldexpf(1.0f + i / 64.0f, -10 - i / 2);  // 2^-10 .. 2^-25: normal, subnormal, below the smallest subnormal
  float *da, *db, *dout; unsigned* dc;
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dout, sizeof ho); hipMalloc(&dc, sizeof hc);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(da, db, dout, dc);
  hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
  for (int i = 0; i < 32; ++i)
    printf("x = %.9e  cvt_pkrtz bits 0x%04x  mfma(x * 1) = %.9e  ratio %.6f\n", ha[i], hc[i], ho[i], ho[i] / ha[i]);
  // subnormal B too: 2^-20 * 2^-4 etc.
  hb[0] = ldexpf(1.0f, -20);
  for (int i = 0; i < 32; ++i) ha[i] = ldexpf(1.0f, -i);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(da, db, dout, dc);
  hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
  for (int i = 0; i < 32; i += 4) printf("2^-%d * 2^-20 = %.9e (exact %.9e)\n", i, ho[i], ldexpf(1.0f, -i - 20));
  return 0;
}
