// valu_mix.hip -- issue cost of mixed instruction streams (2-cycle simple ops next to 4-cycle ops and SALU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void k0(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\tv_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\tv_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\tv_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\tv_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\tv_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4", "scc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k1(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_alignbit_b32 %0, %0, %8, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_alignbit_b32 %4, %4, %8, 31\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_alignbit_b32 %0, %0, %8, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_alignbit_b32 %4, %4, %8, 31\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_alignbit_b32 %0, %0, %8, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_alignbit_b32 %4, %4, %8, 31\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_alignbit_b32 %0, %0, %8, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_alignbit_b32 %4, %4, %8, 31\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_alignbit_b32 %0, %0, %8, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_alignbit_b32 %4, %4, %8, 31\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_alignbit_b32 %0, %0, %8, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_alignbit_b32 %4, %4, %8, 31\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_alignbit_b32 %7, %7, %8, 31\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4", "scc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k2(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_add_u32 %0, %0, %8\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_add_u32 %2, %2, %8\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_add_u32 %4, %4, %8\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_add_u32 %6, %6, %8\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_add_u32 %0, %0, %8\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_add_u32 %2, %2, %8\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_add_u32 %4, %4, %8\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_add_u32 %6, %6, %8\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_add_u32 %0, %0, %8\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_add_u32 %2, %2, %8\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_add_u32 %4, %4, %8\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_add_u32 %6, %6, %8\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_add_u32 %0, %0, %8\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_add_u32 %2, %2, %8\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_add_u32 %4, %4, %8\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_add_u32 %6, %6, %8\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_add_u32 %0, %0, %8\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_add_u32 %2, %2, %8\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_add_u32 %4, %4, %8\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_add_u32 %6, %6, %8\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_add_u32 %0, %0, %8\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_add_u32 %2, %2, %8\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_add_u32 %4, %4, %8\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_add_u32 %6, %6, %8\n\tv_alignbit_b32 %7, %7, %8, 31\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4", "scc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k3(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_add_u32 %3, %3, %8\n\tv_xor_b32 %4, %4, %8\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_add_u32 %6, %6, %8\n\tv_xor_b32 %7, %7, %8\n\tv_alignbit_b32 %0, %0, %8, 31\n\tv_add_u32 %1, %1, %8\n\tv_xor_b32 %2, %2, %8\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_add_u32 %7, %7, %8\n\tv_xor_b32 %0, %0, %8\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_add_u32 %2, %2, %8\n\tv_xor_b32 %3, %3, %8\n\tv_alignbit_b32 %4, %4, %8, 31\n\tv_add_u32 %5, %5, %8\n\tv_xor_b32 %6, %6, %8\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_add_u32 %3, %3, %8\n\tv_xor_b32 %4, %4, %8\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_add_u32 %6, %6, %8\n\tv_xor_b32 %7, %7, %8\n\tv_alignbit_b32 %0, %0, %8, 31\n\tv_add_u32 %1, %1, %8\n\tv_xor_b32 %2, %2, %8\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_add_u32 %7, %7, %8\n\tv_xor_b32 %0, %0, %8\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_add_u32 %2, %2, %8\n\tv_xor_b32 %3, %3, %8\n\tv_alignbit_b32 %4, %4, %8, 31\n\tv_add_u32 %5, %5, %8\n\tv_xor_b32 %6, %6, %8\n\tv_alignbit_b32 %7, %7, %8, 31\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4", "scc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k4(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_alignbit_b32 %7, %7, %8, 31\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4", "scc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k5(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_bitop3_b32 %3, %3, %8, %8 bitop3:0xde\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_bitop3_b32 %7, %7, %8, %8 bitop3:0xde\n\tv_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_bitop3_b32 %3, %3, %8, %8 bitop3:0xde\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_bitop3_b32 %7, %7, %8, %8 bitop3:0xde\n\tv_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_bitop3_b32 %3, %3, %8, %8 bitop3:0xde\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_bitop3_b32 %7, %7, %8, %8 bitop3:0xde\n\tv_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_bitop3_b32 %3, %3, %8, %8 bitop3:0xde\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_bitop3_b32 %7, %7, %8, %8 bitop3:0xde\n\tv_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_bitop3_b32 %3, %3, %8, %8 bitop3:0xde\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_bitop3_b32 %7, %7, %8, %8 bitop3:0xde\n\tv_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_bitop3_b32 %3, %3, %8, %8 bitop3:0xde\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_bitop3_b32 %7, %7, %8, %8 bitop3:0xde\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4", "scc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k6(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_add_u32 %0, %0, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %1, %1, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %2, %2, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %3, %3, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %4, %4, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %5, %5, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %6, %6, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %7, %7, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %0, %0, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %1, %1, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %2, %2, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %3, %3, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %4, %4, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %5, %5, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %6, %6, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %7, %7, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %0, %0, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %1, %1, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %2, %2, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %3, %3, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %4, %4, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %5, %5, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %6, %6, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %7, %7, %8\n\ts_add_i32 s4, s4, 1\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4", "scc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k7(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_add_u32 %0, %0, %8\n\ts_nop 0\n\tv_add_u32 %1, %1, %8\n\ts_nop 0\n\tv_add_u32 %2, %2, %8\n\ts_nop 0\n\tv_add_u32 %3, %3, %8\n\ts_nop 0\n\tv_add_u32 %4, %4, %8\n\ts_nop 0\n\tv_add_u32 %5, %5, %8\n\ts_nop 0\n\tv_add_u32 %6, %6, %8\n\ts_nop 0\n\tv_add_u32 %7, %7, %8\n\ts_nop 0\n\tv_add_u32 %0, %0, %8\n\ts_nop 0\n\tv_add_u32 %1, %1, %8\n\ts_nop 0\n\tv_add_u32 %2, %2, %8\n\ts_nop 0\n\tv_add_u32 %3, %3, %8\n\ts_nop 0\n\tv_add_u32 %4, %4, %8\n\ts_nop 0\n\tv_add_u32 %5, %5, %8\n\ts_nop 0\n\tv_add_u32 %6, %6, %8\n\ts_nop 0\n\tv_add_u32 %7, %7, %8\n\ts_nop 0\n\tv_add_u32 %0, %0, %8\n\ts_nop 0\n\tv_add_u32 %1, %1, %8\n\ts_nop 0\n\tv_add_u32 %2, %2, %8\n\ts_nop 0\n\tv_add_u32 %3, %3, %8\n\ts_nop 0\n\tv_add_u32 %4, %4, %8\n\ts_nop 0\n\tv_add_u32 %5, %5, %8\n\ts_nop 0\n\tv_add_u32 %6, %6, %8\n\ts_nop 0\n\tv_add_u32 %7, %7, %8\n\ts_nop 0\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4", "scc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k8(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_or_b32 %3, %3, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_or_b32 %7, %7, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_or_b32 %3, %3, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_or_b32 %7, %7, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_or_b32 %3, %3, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_or_b32 %7, %7, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_or_b32 %3, %3, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_or_b32 %7, %7, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_or_b32 %3, %3, %8\n\ts_add_i32 s4, s4, 1\n\tv_add_u32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_or_b32 %7, %7, %8\n\ts_add_i32 s4, s4, 1\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4", "scc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
typedef void (*kern)(uint32_t *, int, uint32_t);
static void run(const char *name, kern k, uint32_t *d, int nv) {
    const int blocks = 8192, iters = 2048;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // cycles per VALU instruction per SIMD at 8 waves / SIMD, assuming 2.4 GHz
    const double valu = (double)blocks * 4 * iters * nv;            // wave-instructions
    printf("%-10s %.3f ms  %.2f SIMD-cycles per VALU instruction (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / valu);
}
int main() {
    uint32_t *d;
    (void)hipMalloc(&d, 8192 * 256 * 4);
    run("A", k0, d, 48);
    run("S", k1, d, 48);
    run("AS", k2, d, 48);
    run("AAS", k3, d, 48);
    run("AAAS", k4, d, 48);
    run("AASS", k5, d, 48);
    run("A+salu", k6, d, 24);
    run("A+nop", k7, d, 24);
    run("AAAA+salu", k8, d, 40);
    return 0;
}
