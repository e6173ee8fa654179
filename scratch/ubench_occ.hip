// How many waves per SIMD does the v_mad_u64_u32 MAC block need to saturate the VALU?
// Kernel: NACC independent 64-bit accumulators per lane, each MAD uses an SGPR multiplier (like k_matvec3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
constexpr int ITER = 4096;
template<int NACC, int PAD> __global__ void __launch_bounds__(256) k(uint32_t* out, const uint32_t* __restrict__ m, int iters) {
  __shared__ uint32_t lds[PAD > 0 ? PAD : 1];
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  uint64_t acc[NACC]; uint32_t x[9];
  for (int i = 0; i < NACC; i++) acc[i] = t + i;
  for (int i = 0; i < 9; i++) x[i] = t * 7 + i;
  if (PAD > 0 && threadIdx.x == 0) lds[0] = t;
  for (int it = 0; it < iters; it++) {
    uint32_t m0 = m[(it * 4) & 1023], m1 = m[(it * 4 + 1) & 1023], m2 = m[(it * 4 + 2) & 1023], m3 = m[(it * 4 + 3) & 1023];
#pragma unroll
    for (int j = 0; j < 9; j++) { acc[(j) % NACC] += (uint64_t)m0 * x[j]; }
#pragma unroll
    for (int j = 0; j < 9; j++) { acc[(j + 9) % NACC] += (uint64_t)m1 * x[j]; }
#pragma unroll
    for (int j = 0; j < 9; j++) { acc[(j + 18) % NACC] += (uint64_t)m2 * x[j]; }
#pragma unroll
    for (int j = 0; j < 9; j++) { acc[(j + 27) % NACC] += (uint64_t)m3 * x[j]; }
  }
  uint64_t r = 0; for (int i = 0; i < NACC; i++) r += acc[i];
  out[t] = (uint32_t)r + (uint32_t)(r >> 32) + (PAD > 0 ? lds[0] : 0);
}
template<int NACC, int PAD> int run(int blocks_per_cu, const uint32_t* m, uint32_t* d) {
  int blocks = 256 * blocks_per_cu;
  k<NACC, PAD><<<blocks, 256>>>(d, m, 64); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0)); for (int r = 0; r < 3; r++) k<NACC, PAD><<<blocks, 256>>>(d, m, ITER); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  double mads_per_simd = (double)blocks_per_cu * ITER * 36;   // one wave per SIMD per block
  double cyc = ms * 1e-3 * 2.35e9 / mads_per_simd;
  printf("NACC=%2d waves/SIMD=%d (LDS pad %5d B): %7.3f ms  -> %5.2f cycles per MAD per SIMD @2.35GHz  (%.1f T MAD/s)\n", NACC, blocks_per_cu, PAD * 4, ms, cyc, 1024.0 * mads_per_simd * 64 / ms * 1e-9);
  return 0;
}
int main() {
  uint32_t *m, *d; CK(hipMalloc(&m, 4096)); CK(hipMemset(m, 0x5a, 4096)); CK(hipMalloc(&d, 256 * 8 * 256 * 4));
  // occupancy is forced by LDS padding: 160 KB / pad => blocks per CU
  run<36, 36 * 1024>(1, m, d); run<36, 18 * 1024>(2, m, d); run<36, 12 * 1024>(3, m, d); run<36, 9 * 1024>(4, m, d); run<36, 0>(5, m, d); run<36, 0>(6, m, d); run<36, 0>(8, m, d);
  run<9, 36 * 1024>(1, m, d); run<9, 18 * 1024>(2, m, d); run<9, 12 * 1024>(3, m, d); run<9, 9 * 1024>(4, m, d); run<9, 0>(8, m, d);
  run<18, 36 * 1024>(1, m, d); run<18, 12 * 1024>(3, m, d);
  return 0;
}
