// MFMA issue-rate microbenchmark (gfx950): cycles per instruction per SIMD for the integer matrix instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
constexpr int ITER = 4096;
template <int OP> __global__ void __launch_bounds__(256, 1) k(int *out, const v4i *a, const v4i *b) {
    v4i av = a[threadIdx.x & 63], bv = b[threadIdx.x & 63];
    if constexpr (OP == 0) {
        v4i c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        for (int it = 0; it < ITER; it++) {
            asm volatile("v_mfma_i32_16x16x64_i8 %0, %8, %9, %0\n v_mfma_i32_16x16x64_i8 %1, %8, %9, %1\n v_mfma_i32_16x16x64_i8 %2, %8, %9, %2\n v_mfma_i32_16x16x64_i8 %3, %8, %9, %3\n"
                         "v_mfma_i32_16x16x64_i8 %4, %8, %9, %4\n v_mfma_i32_16x16x64_i8 %5, %8, %9, %5\n v_mfma_i32_16x16x64_i8 %6, %8, %9, %6\n v_mfma_i32_16x16x64_i8 %7, %8, %9, %7\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(av), "v"(bv));
        }
        out[threadIdx.x + blockIdx.x * 256] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
    } else if constexpr (OP == 1) {
        v16i c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int it = 0; it < ITER; it++) {
            asm volatile("v_mfma_i32_32x32x32_i8 %0, %4, %5, %0\n v_mfma_i32_32x32x32_i8 %1, %4, %5, %1\n v_mfma_i32_32x32x32_i8 %2, %4, %5, %2\n v_mfma_i32_32x32x32_i8 %3, %4, %5, %3\n"
                         "v_mfma_i32_32x32x32_i8 %0, %4, %5, %0\n v_mfma_i32_32x32x32_i8 %1, %4, %5, %1\n v_mfma_i32_32x32x32_i8 %2, %4, %5, %2\n v_mfma_i32_32x32x32_i8 %3, %4, %5, %3\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv));
        }
        out[threadIdx.x + blockIdx.x * 256] = c0[0] + c1[1] + c2[2] + c3[3];
    } else if constexpr (OP == 2) {   // legacy 16x16x32 i8 (64-bit operands)
        typedef int v2i __attribute__((ext_vector_type(2)));
        v2i a2 = {av[0], av[1]}, b2 = {bv[0], bv[1]};
        v4i c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        for (int it = 0; it < ITER; it++) {
            asm volatile("v_mfma_i32_16x16x32_i8 %0, %8, %9, %0\n v_mfma_i32_16x16x32_i8 %1, %8, %9, %1\n v_mfma_i32_16x16x32_i8 %2, %8, %9, %2\n v_mfma_i32_16x16x32_i8 %3, %8, %9, %3\n"
                         "v_mfma_i32_16x16x32_i8 %4, %8, %9, %4\n v_mfma_i32_16x16x32_i8 %5, %8, %9, %5\n v_mfma_i32_16x16x32_i8 %6, %8, %9, %6\n v_mfma_i32_16x16x32_i8 %7, %8, %9, %7\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a2), "v"(b2));
        }
        out[threadIdx.x + blockIdx.x * 256] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
    } else if constexpr (OP == 3) {   // 16x16x64 i8 interleaved with 2 full-rate + 1 half-rate VALU per MFMA (same wave)
        v4i c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        uint32_t x = threadIdx.x, y = blockIdx.x, z = 7;
        for (int it = 0; it < ITER; it++) {
#define ONE(C) "v_mfma_i32_16x16x64_i8 " C ", %11, %12, " C "\n v_xor_b32 %8, %9, %8\n v_alignbyte_b32 %9, %8, %10, 1\n v_mov_b32 %10, %9\n"
            asm volatile(ONE("%0") ONE("%1") ONE("%2") ONE("%3") ONE("%4") ONE("%5") ONE("%6") ONE("%7")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), "+v"(x), "+v"(y), "+v"(z) : "v"(av), "v"(bv));
        }
        out[threadIdx.x + blockIdx.x * 256] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3] + x + y + z;
    }
}
template <int OP> int run(const char *name, double macs_per_instr, int per_iter) {
    int *d; v4i *a, *b; int blocks = 256 * 4;
    CK(hipMalloc(&d, blocks * 256 * 4)); CK(hipMalloc(&a, 64 * 16)); CK(hipMalloc(&b, 64 * 16));
    CK(hipMemset(a, 1, 64 * 16)); CK(hipMemset(b, 1, 64 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<OP><<<blocks, 256>>>(d, a, b); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) k<OP><<<blocks, 256>>>(d, a, b); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    // waves per SIMD in sequence: blocks*4 waves over 1024 SIMDs, one wave per SIMD at a time (launch_bounds(256,1) does not force it, but
    // the count below is per SIMD regardless of residency)
    double instr_per_simd = (double)blocks * 4 / 1024.0 * ITER * per_iter;
    double ns_per = ms * 1e6 / instr_per_simd;
    printf("%-44s %8.3f ms  %6.2f ns per MFMA per SIMD (= %5.1f cycles @2.4GHz, %5.1f @2.0GHz)  %7.1f T MAC/s\n", name, ms, ns_per, ns_per * 2.4, ns_per * 2.0,
           (double)blocks * 4 * ITER * per_iter * macs_per_instr / (ms * 1e-3) / 1e12);
    return 0;
}
int main() {
    run<0>("v_mfma_i32_16x16x64_i8 (8 independent)", 16.0 * 16 * 64, 8);
    run<1>("v_mfma_i32_32x32x32_i8 (4 independent x2)", 32.0 * 32 * 32, 8);
    run<2>("v_mfma_i32_16x16x32_i8 (8 independent)", 16.0 * 16 * 32, 8);
    run<3>("16x16x64_i8 + xor + alignbyte + mov each", 16.0 * 16 * 64, 8);
    return 0;
}
