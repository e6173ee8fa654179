// Is s_memtime the shader clock, and what does that clock do under load?  One wave per CU runs a dependent
// v_add chain (light), then the same chain while other blocks keep the matrix cores and the VALU busy (heavy).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned long long *out, int iters, int heavy_blocks) {
    if ((int)blockIdx.x >= heavy_blocks) {              // measuring wave
        if (threadIdx.x >= 64) return;
        unsigned v = threadIdx.x;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(v));
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) { out[2 * (blockIdx.x - heavy_blocks)] = t1 - t0; out[2 * (blockIdx.x - heavy_blocks) + 1] = v; }
        return;
    }
    // heavy: MFMA + MAD stream
    v4i acc = {0, 0, 0, 0}, a = {1, 2, 3, 4}, b = {5, 6, 7, 8};
    unsigned long long m = threadIdx.x;
    for (int i = 0; i < iters * 6; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
            m = m * 0x9e3779b9u + (unsigned)acc[0];
            m = m * 0x85ebca6bu + 1;
        }
    }
    if (m == 42 && acc[1] == 7) out[0] = 1;
}
int main() {
    unsigned long long *d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int heavy = 0; heavy <= 1; heavy++) {
        const int hb = heavy ? 256 * 7 : 0, mb = 8;
        for (int rep = 0; rep < 3; rep++) {
            hipMemset(d, 0, 4096);
            hipEventRecord(e0);
            probe<<<hb + mb, 256>>>(d, iters, hb);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("%s rep %d: kernel %.3f ms; chain of %d dependent v_add: %llu ticks = %.3f ticks/add; ticks per us of kernel time >= %.0f\n",
                   heavy ? "heavy" : "light", rep, ms, iters * 64, h[0], (double)h[0] / (iters * 64.0), (double)h[0] / (ms * 1e3));
        }
    }
    return 0;
}
