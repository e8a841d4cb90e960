// valu_dep.hip -- does the 2-cycle issue of the simple VALU ops need independent neighbours?
// CHAINS independent dependency chains per wave, interleaved round-robin; 32 ops per iteration.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int CHAINS, int OPK>
__global__ __launch_bounds__(256) void spin(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    uint32_t b = seed | 1u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            uint32_t &x = a[r % CHAINS];
            if (OPK == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
            else if (OPK == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
            else if (OPK == 2) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0xde" : "+v"(x) : "v"(b));
            else asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(x) : "v"(b));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int CHAINS, int OPK>
static void run(const char *name, uint32_t *d) {
    const int blocks = 8192, iters = 2048;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((spin<CHAINS, OPK>), dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((spin<CHAINS, OPK>), dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-12s chains=%d  %.3f ms  %6.2f T lane-ops/s\n", name, CHAINS, ms, (double)blocks * 256 * iters * 32 / (ms * 1e-3) / 1e12);
}

int main() {
    uint32_t *d;
    (void)hipMalloc(&d, 8192 * 256 * 4);
    run<1, 0>("v_add_u32", d); run<2, 0>("v_add_u32", d); run<4, 0>("v_add_u32", d); run<8, 0>("v_add_u32", d);
    run<1, 1>("v_xor_b32", d); run<2, 1>("v_xor_b32", d); run<4, 1>("v_xor_b32", d);
    run<1, 2>("v_bitop3", d); run<2, 2>("v_bitop3", d); run<4, 2>("v_bitop3", d);
    run<1, 3>("v_alignbit", d); run<2, 3>("v_alignbit", d);
    return 0;
}
