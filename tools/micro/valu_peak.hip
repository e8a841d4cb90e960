// valu_peak.hip -- integer VALU issue rate of one MI355X, measured: every wave runs a loop of
// 32 independent 32-bit VALU ops per iteration (8 chains x 4); prints lane-ops/s for plain
// v_add_u32, for v_bitop3_b32 and for v_min3_u32.  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define OPS8(OP) \
    OP " %0, %0, %8\n\t" OP " %1, %1, %8\n\t" OP " %2, %2, %8\n\t" OP " %3, %3, %8\n\t" \
    OP " %4, %4, %8\n\t" OP " %5, %5, %8\n\t" OP " %6, %6, %8\n\t" OP " %7, %7, %8\n\t"

template <int KIND>
__global__ __launch_bounds__(256) void spin(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0)
            asm volatile(OPS8("v_add_u32") OPS8("v_add_u32") OPS8("v_add_u32") OPS8("v_add_u32")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        else if (KIND == 1)
            asm volatile(OPS8("v_xor_b32") OPS8("v_xor_b32") OPS8("v_xor_b32") OPS8("v_xor_b32")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        else
            asm volatile(OPS8("v_min_u32") OPS8("v_min_u32") OPS8("v_min_u32") OPS8("v_min_u32")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int KIND>
static void run(const char *name, uint32_t *d, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * (double)iters * 32;
    printf("%-10s blocks=%d  %.3f ms  %.2f T lane-ops/s\n", name, blocks, ms, ops / (ms * 1e-3) / 1e12);
}

int main() {
    uint32_t *d;
    hipMalloc(&d, 8192 * 256 * 4);
    for (int blocks : {2048, 4096, 8192}) {
        run<0>("v_add_u32", d, blocks, 4096);
        run<1>("v_xor_b32", d, blocks, 4096);
        run<2>("v_min_u32", d, blocks, 4096);
    }
    return 0;
}
