// What a tile load costs a wave that is alone on its SIMD (k_mm8w): the LDS-DMA form (global_load_lds_dwordx4, 1 KB per instruction)
// against a plain global_load_dwordx4 into registers followed, four loads later, by a ds_write_b128 -- with S filler VALU instructions
// (v_mad_u64_u32 on independent accumulators) between consecutive loads, from sources that miss every cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
constexpr int ITER = 256;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <int MODE, int S> __global__ void __launch_bounds__(256, 1) k(int *out, const uint4 *src) {
    extern __shared__ uint4 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t m[8];
    for (int i = 0; i < 8; i++) m[i] = threadIdx.x * 7 + i;
    uint32_t x = threadIdx.x * 2654435761u + 1;
    const uint4 *p = src + ((size_t)(blockIdx.x * 4 + wave) * ITER) * 64 + lane;
    uint4 *ring = lds + wave * 8 * 64;          // eight 1 KB slots per wave
    v4u r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0;
    for (int it = 0; it < ITER; it++) {
        if constexpr (MODE == 0) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(ring + (it & 7) * 64));
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(p + (size_t)it * 64), "s"(dst) : "memory");
        } else if constexpr (MODE == 1) {
            // the set loaded four iterations ago goes to LDS, then is reloaded
            const uint32_t dsta = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(ring + (it & 7) * 64 + lane);
            switch (it & 3) {
#define STEP(R) asm volatile("s_waitcnt vmcnt(3)\n\tds_write_b128 %1, %0\n\tglobal_load_dwordx4 %0, %2, off" : "+v"(R) : "v"(dsta), "v"(p + (size_t)it * 64) : "memory")
                case 0: STEP(r0); break;
                case 1: STEP(r1); break;
                case 2: STEP(r2); break;
                default: STEP(r3); break;
            }
        }
#pragma unroll
        for (int s = 0; s < S; s++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(m[s & 7]) : "v"(x) : "vcc");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    uint64_t t = 0;
    for (int i = 0; i < 8; i++) t += m[i];
    out[threadIdx.x + blockIdx.x * 256] = (int)t + (int)lds[threadIdx.x].x + (int)(r0[0] + r1[1] + r2[2] + r3[3]);
}
template <int MODE, int S> int run(const char *name, const uint4 *src, int *d) {
    const int blocks = 256;
    const size_t ldsb = 150 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, S>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE, S><<<blocks, 256, ldsb>>>(d, src); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) k<MODE, S><<<blocks, 256, ldsb>>>(d, src); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-28s S=%4d: %8.1f us per launch, %7.1f ns per iteration per wave (%6.0f cycles @2.0 GHz)\n", name, S, ms * 1e3, ms * 1e6 / ITER, ms * 1e6 / ITER * 2.0);
    return 0;
}
int main() {
    uint4 *src; int *d;
    const size_t bytes = (size_t)256 * 4 * ITER * 1024;          // 256 MB: every load misses
    CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes)); CK(hipMalloc(&d, 256 * 256 * 4));
#define ROW(S) run<2, S>("VALU filler only", src, d); run<0, S>("LDS-DMA + filler", src, d); run<1, S>("load + ds_write + filler", src, d);
    ROW(0) ROW(16) ROW(64) ROW(256) ROW(1024)
    return 0;
}
