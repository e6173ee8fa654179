// Instruction-rate microbenchmark for gfx950 integer paths (design input for the Fp kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
constexpr int ITER = 65536;
constexpr int UNR = 8;

template<int OP> __global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed, uint32_t sa) {
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  uint64_t acc[UNR]; uint32_t a[UNR], b[UNR]; double fa[UNR], fb[UNR], fc[UNR];
  for (int i = 0; i < UNR; i++) { acc[i] = t * 77 + i + seed; a[i] = t * 3 + i * 5 + seed; b[i] = t + i * 7 + 1; fa[i] = a[i]; fb[i] = 1.0 + 1e-9 * b[i]; fc[i] = i; }
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < UNR; i++) {
      if constexpr (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
      if constexpr (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 4) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b[i]) : "vcc");
      if constexpr (OP == 5) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(fc[i]) : "v"(fa[i]), "v"(fb[i]));
      if constexpr (OP == 6) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 8) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "s"(sa), "v"(b[i]) : "vcc");
      if constexpr (OP == 9) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(b[(i+1)%UNR]), "v"(a[(i+1)%UNR]) : "vcc");
      if constexpr (OP == 10) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(acc[i]), "+v"(a[i]) : "v"(a[(i+1)%UNR]), "v"(b[i]) : "vcc");
      if constexpr (OP == 11) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 12) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 13) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i+1)%UNR]));
      if constexpr (OP == 14) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 16) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(acc[i]));
      if constexpr (OP == 17) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 18) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(a[i]));
      if constexpr (OP == 19) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i+1)%UNR]));
      if constexpr (OP == 20) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(a[i]));
      if constexpr (OP == 21) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %3, vcc, %1, %4, %3" : "+v"(acc[i]), "+v"(a[0]) : "v"(b[i]), "v"(acc[(i+1)%UNR]), "v"(b[(i+1)%UNR]) : "vcc");
      if constexpr (OP == 15) asm volatile("v_mad_u64_u32 %0, %3, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i]), "s"((uint64_t)0) : );
    }
  }
  uint32_t r = 0; for (int i = 0; i < UNR; i++) r += (uint32_t)acc[i] + (uint32_t)(acc[i] >> 32) + a[i] + b[i] + (uint32_t)fc[i];
  out[t] = r;
}
template<int OP> int run(const char* name, int nops, int wpb) {
  uint32_t* d; int blocks = 256 * 8; CK(hipMalloc(&d, blocks * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<OP><<<blocks, wpb>>>(d, 1, 12345); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) k<OP><<<blocks, wpb>>>(d, r, 12345 + r); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  double ops = (double)blocks * wpb * ITER * UNR * nops;
  // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs
  double wave_instr = ops / 64.0; double cyc = ms * 1e-3 * 2.4e9 * 1024 / wave_instr;
  printf("%-34s %8.3f ms  %9.2f Gop/s  ~%5.2f cyc/wave-instr/SIMD@2.4GHz (threads/block %d)\n", name, ms, ops / ms * 1e-6, cyc, wpb);
  CK(hipFree(d)); return 0;
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  for (int wpb : {256}) {
    run<7>("v_add_u32", 1, wpb); run<12>("v_lshl_add_u32", 1, wpb); run<13>("v_fma_f32", 1, wpb);
    run<4>("v_add_co_u32", 1, wpb); run<9>("v_add_co+v_addc (pair)", 2, wpb);
    run<0>("v_mad_u64_u32 (vcc)", 1, wpb); run<15>("v_mad_u64_u32 (sgpr carry)", 1, wpb); run<8>("v_mad_u64_u32 sgpr src", 1, wpb); run<10>("v_mad_u64_u32+v_addc (pair)", 2, wpb);
    run<1>("v_mul_lo_u32", 1, wpb); run<2>("v_mul_hi_u32", 1, wpb);
    run<3>("v_mad_u32_u24", 1, wpb); run<14>("v_mad_i32_i24", 1, wpb); run<11>("v_mul_u32_u24", 1, wpb); run<6>("v_mul_hi_u32_u24", 1, wpb);
    run<5>("v_fma_f64", 1, wpb);
    run<16>("v_lshrrev_b64", 1, wpb); run<17>("v_alignbit_b32", 1, wpb); run<18>("v_and_b32", 1, wpb); run<19>("v_lshl_add_u64", 1, wpb); run<20>("v_bfe_u32", 1, wpb);
  }
  return 0;
}
