// Issue rate of the instruction mix of a k_mm8w pass (1 int8 MFMA : 2 v_mad_u64_u32 : 1 v_alignbit : 2 full-rate VALU) at 1, 2, 3, 4
// waves per SIMD (occupancy set by an LDS pad): what a second resident wave would buy that kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
constexpr int ITER = 2048;
template <int MODE> __global__ void __launch_bounds__(256) k(int *out, const v4i *a, const v4i *b) {
    extern __shared__ int pad[];
    v4i av = a[threadIdx.x & 63], bv = b[threadIdx.x & 63];
    v4i c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    uint64_t m0 = threadIdx.x, m1 = blockIdx.x, m2 = 3, m3 = 5, m4 = 7, m5 = 9, m6 = 11, m7 = 13;
    uint32_t x = threadIdx.x * 2654435761u, y = blockIdx.x + 1, z = 7, w = 9;
    if (threadIdx.x == 1023) pad[0] = 1;
    for (int it = 0; it < ITER; it++) {
#define VALU6(MA, MB) " v_mad_u64_u32 " MA ", vcc, %12, %13, " MA "\n v_alignbit_b32 %14, %12, %13, 7\n v_and_b32 %15, 0x1fffffff, %14\n v_mad_u64_u32 " MB ", vcc, %13, %15, " MB "\n v_mov_b32 %12, %15\n v_xor_b32 %13, %14, %13\n"
#define SLOT(C, MA, MB) "v_mfma_i32_16x16x64_i8 " C ", %16, %17, " C "\n" VALU6(MA, MB)
#define SLOTV(C, MA, MB) VALU6(MA, MB)
        if constexpr (MODE == 0)
            asm volatile(SLOT("%0", "%4", "%5") SLOT("%1", "%6", "%7") SLOT("%2", "%8", "%9") SLOT("%3", "%10", "%11")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "+v"(m6), "+v"(m7),
                           "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(av), "v"(bv) : "vcc");
        else
            asm volatile(SLOTV("%0", "%4", "%5") SLOTV("%1", "%6", "%7") SLOTV("%2", "%8", "%9") SLOTV("%3", "%10", "%11")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "+v"(m6), "+v"(m7),
                           "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(av), "v"(bv) : "vcc");
    }
    out[threadIdx.x + blockIdx.x * 256] = c0[0] + c1[1] + c2[2] + c3[3] + (int)(m0 + m1 + m2 + m3 + m4 + m5 + m6 + m7) + x + y + z + w;
}
template <int MODE> int run(const char *name, int waves_per_simd) {
    int *d; v4i *a, *b;
    const int blocks = 256 * waves_per_simd * 4;                 // four rounds of full residency
    const size_t lds = (size_t)(160 * 1024 / waves_per_simd) - 1024;
    CK(hipMalloc(&d, (size_t)blocks * 256 * 4)); CK(hipMalloc(&a, 64 * 16)); CK(hipMalloc(&b, 64 * 16));
    CK(hipMemset(a, 1, 64 * 16)); CK(hipMemset(b, 1, 64 * 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE><<<blocks, 256, lds>>>(d, a, b); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) k<MODE><<<blocks, 256, lds>>>(d, a, b); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double slots_per_simd = (double)blocks * 4 / 1024.0 * ITER * 4;
    const double ns = ms * 1e6 / slots_per_simd;
    printf("%-34s %d wave(s) per SIMD: %7.3f ms  %6.2f ns per slot per SIMD (%5.1f cycles @2.0 GHz; %4.1f per instruction)\n", name, waves_per_simd, ms, ns, ns * 2.0,
           ns * 2.0 / (MODE == 0 ? 7 : 6));
    CK(hipFree(d)); CK(hipFree(a)); CK(hipFree(b));
    return 0;
}
int main() {
    for (int w = 1; w <= 4; w++) run<0>("MFMA + 6 VALU per slot", w);
    for (int w = 1; w <= 4; w++) run<1>("6 VALU per slot (no MFMA)", w);
    return 0;
}
