// valu_rates.hip -- issue rate of the 32-bit integer VALU ops the kernels use, measured on one
// MI355X: every wave runs 32 independent ops (8 chains x 4) per loop iteration.
// One kernel per op (machine-written from a list of op templates; the asm bodies are 32 copies of one line).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void k0(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\tv_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\tv_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\tv_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k1(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_xor_b32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_xor_b32 %2, %2, %8\n\tv_xor_b32 %3, %3, %8\n\tv_xor_b32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_xor_b32 %6, %6, %8\n\tv_xor_b32 %7, %7, %8\n\tv_xor_b32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_xor_b32 %2, %2, %8\n\tv_xor_b32 %3, %3, %8\n\tv_xor_b32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_xor_b32 %6, %6, %8\n\tv_xor_b32 %7, %7, %8\n\tv_xor_b32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_xor_b32 %2, %2, %8\n\tv_xor_b32 %3, %3, %8\n\tv_xor_b32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_xor_b32 %6, %6, %8\n\tv_xor_b32 %7, %7, %8\n\tv_xor_b32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_xor_b32 %2, %2, %8\n\tv_xor_b32 %3, %3, %8\n\tv_xor_b32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_xor_b32 %6, %6, %8\n\tv_xor_b32 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k2(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_and_b32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_and_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8\n\tv_and_b32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_and_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8\n\tv_and_b32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_and_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8\n\tv_and_b32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_and_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k3(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_or_b32 %0, %0, %8\n\tv_or_b32 %1, %1, %8\n\tv_or_b32 %2, %2, %8\n\tv_or_b32 %3, %3, %8\n\tv_or_b32 %4, %4, %8\n\tv_or_b32 %5, %5, %8\n\tv_or_b32 %6, %6, %8\n\tv_or_b32 %7, %7, %8\n\tv_or_b32 %0, %0, %8\n\tv_or_b32 %1, %1, %8\n\tv_or_b32 %2, %2, %8\n\tv_or_b32 %3, %3, %8\n\tv_or_b32 %4, %4, %8\n\tv_or_b32 %5, %5, %8\n\tv_or_b32 %6, %6, %8\n\tv_or_b32 %7, %7, %8\n\tv_or_b32 %0, %0, %8\n\tv_or_b32 %1, %1, %8\n\tv_or_b32 %2, %2, %8\n\tv_or_b32 %3, %3, %8\n\tv_or_b32 %4, %4, %8\n\tv_or_b32 %5, %5, %8\n\tv_or_b32 %6, %6, %8\n\tv_or_b32 %7, %7, %8\n\tv_or_b32 %0, %0, %8\n\tv_or_b32 %1, %1, %8\n\tv_or_b32 %2, %2, %8\n\tv_or_b32 %3, %3, %8\n\tv_or_b32 %4, %4, %8\n\tv_or_b32 %5, %5, %8\n\tv_or_b32 %6, %6, %8\n\tv_or_b32 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k4(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_min_u32 %0, %0, %8\n\tv_min_u32 %1, %1, %8\n\tv_min_u32 %2, %2, %8\n\tv_min_u32 %3, %3, %8\n\tv_min_u32 %4, %4, %8\n\tv_min_u32 %5, %5, %8\n\tv_min_u32 %6, %6, %8\n\tv_min_u32 %7, %7, %8\n\tv_min_u32 %0, %0, %8\n\tv_min_u32 %1, %1, %8\n\tv_min_u32 %2, %2, %8\n\tv_min_u32 %3, %3, %8\n\tv_min_u32 %4, %4, %8\n\tv_min_u32 %5, %5, %8\n\tv_min_u32 %6, %6, %8\n\tv_min_u32 %7, %7, %8\n\tv_min_u32 %0, %0, %8\n\tv_min_u32 %1, %1, %8\n\tv_min_u32 %2, %2, %8\n\tv_min_u32 %3, %3, %8\n\tv_min_u32 %4, %4, %8\n\tv_min_u32 %5, %5, %8\n\tv_min_u32 %6, %6, %8\n\tv_min_u32 %7, %7, %8\n\tv_min_u32 %0, %0, %8\n\tv_min_u32 %1, %1, %8\n\tv_min_u32 %2, %2, %8\n\tv_min_u32 %3, %3, %8\n\tv_min_u32 %4, %4, %8\n\tv_min_u32 %5, %5, %8\n\tv_min_u32 %6, %6, %8\n\tv_min_u32 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k5(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_max_u32 %0, %0, %8\n\tv_max_u32 %1, %1, %8\n\tv_max_u32 %2, %2, %8\n\tv_max_u32 %3, %3, %8\n\tv_max_u32 %4, %4, %8\n\tv_max_u32 %5, %5, %8\n\tv_max_u32 %6, %6, %8\n\tv_max_u32 %7, %7, %8\n\tv_max_u32 %0, %0, %8\n\tv_max_u32 %1, %1, %8\n\tv_max_u32 %2, %2, %8\n\tv_max_u32 %3, %3, %8\n\tv_max_u32 %4, %4, %8\n\tv_max_u32 %5, %5, %8\n\tv_max_u32 %6, %6, %8\n\tv_max_u32 %7, %7, %8\n\tv_max_u32 %0, %0, %8\n\tv_max_u32 %1, %1, %8\n\tv_max_u32 %2, %2, %8\n\tv_max_u32 %3, %3, %8\n\tv_max_u32 %4, %4, %8\n\tv_max_u32 %5, %5, %8\n\tv_max_u32 %6, %6, %8\n\tv_max_u32 %7, %7, %8\n\tv_max_u32 %0, %0, %8\n\tv_max_u32 %1, %1, %8\n\tv_max_u32 %2, %2, %8\n\tv_max_u32 %3, %3, %8\n\tv_max_u32 %4, %4, %8\n\tv_max_u32 %5, %5, %8\n\tv_max_u32 %6, %6, %8\n\tv_max_u32 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k6(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_min3_u32 %0, %0, %8, %8\n\tv_min3_u32 %1, %1, %8, %8\n\tv_min3_u32 %2, %2, %8, %8\n\tv_min3_u32 %3, %3, %8, %8\n\tv_min3_u32 %4, %4, %8, %8\n\tv_min3_u32 %5, %5, %8, %8\n\tv_min3_u32 %6, %6, %8, %8\n\tv_min3_u32 %7, %7, %8, %8\n\tv_min3_u32 %0, %0, %8, %8\n\tv_min3_u32 %1, %1, %8, %8\n\tv_min3_u32 %2, %2, %8, %8\n\tv_min3_u32 %3, %3, %8, %8\n\tv_min3_u32 %4, %4, %8, %8\n\tv_min3_u32 %5, %5, %8, %8\n\tv_min3_u32 %6, %6, %8, %8\n\tv_min3_u32 %7, %7, %8, %8\n\tv_min3_u32 %0, %0, %8, %8\n\tv_min3_u32 %1, %1, %8, %8\n\tv_min3_u32 %2, %2, %8, %8\n\tv_min3_u32 %3, %3, %8, %8\n\tv_min3_u32 %4, %4, %8, %8\n\tv_min3_u32 %5, %5, %8, %8\n\tv_min3_u32 %6, %6, %8, %8\n\tv_min3_u32 %7, %7, %8, %8\n\tv_min3_u32 %0, %0, %8, %8\n\tv_min3_u32 %1, %1, %8, %8\n\tv_min3_u32 %2, %2, %8, %8\n\tv_min3_u32 %3, %3, %8, %8\n\tv_min3_u32 %4, %4, %8, %8\n\tv_min3_u32 %5, %5, %8, %8\n\tv_min3_u32 %6, %6, %8, %8\n\tv_min3_u32 %7, %7, %8, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k7(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_lshlrev_b32 %2, 1, %2\n\tv_lshlrev_b32 %3, 1, %3\n\tv_lshlrev_b32 %4, 1, %4\n\tv_lshlrev_b32 %5, 1, %5\n\tv_lshlrev_b32 %6, 1, %6\n\tv_lshlrev_b32 %7, 1, %7\n\tv_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_lshlrev_b32 %2, 1, %2\n\tv_lshlrev_b32 %3, 1, %3\n\tv_lshlrev_b32 %4, 1, %4\n\tv_lshlrev_b32 %5, 1, %5\n\tv_lshlrev_b32 %6, 1, %6\n\tv_lshlrev_b32 %7, 1, %7\n\tv_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_lshlrev_b32 %2, 1, %2\n\tv_lshlrev_b32 %3, 1, %3\n\tv_lshlrev_b32 %4, 1, %4\n\tv_lshlrev_b32 %5, 1, %5\n\tv_lshlrev_b32 %6, 1, %6\n\tv_lshlrev_b32 %7, 1, %7\n\tv_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_lshlrev_b32 %2, 1, %2\n\tv_lshlrev_b32 %3, 1, %3\n\tv_lshlrev_b32 %4, 1, %4\n\tv_lshlrev_b32 %5, 1, %5\n\tv_lshlrev_b32 %6, 1, %6\n\tv_lshlrev_b32 %7, 1, %7\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k8(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %4, 1, %4\n\tv_lshrrev_b32 %5, 1, %5\n\tv_lshrrev_b32 %6, 1, %6\n\tv_lshrrev_b32 %7, 1, %7\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %4, 1, %4\n\tv_lshrrev_b32 %5, 1, %5\n\tv_lshrrev_b32 %6, 1, %6\n\tv_lshrrev_b32 %7, 1, %7\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %4, 1, %4\n\tv_lshrrev_b32 %5, 1, %5\n\tv_lshrrev_b32 %6, 1, %6\n\tv_lshrrev_b32 %7, 1, %7\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %1, 1, %1\n\tv_lshrrev_b32 %2, 1, %2\n\tv_lshrrev_b32 %3, 1, %3\n\tv_lshrrev_b32 %4, 1, %4\n\tv_lshrrev_b32 %5, 1, %5\n\tv_lshrrev_b32 %6, 1, %6\n\tv_lshrrev_b32 %7, 1, %7\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k9(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_bitop3_b32 %0, %0, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %1, %1, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %2, %2, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %3, %3, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %4, %4, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %5, %5, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %6, %6, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %7, %7, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %0, %0, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %1, %1, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %2, %2, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %3, %3, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %4, %4, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %5, %5, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %6, %6, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %7, %7, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %0, %0, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %1, %1, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %2, %2, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %3, %3, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %4, %4, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %5, %5, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %6, %6, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %7, %7, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %0, %0, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %1, %1, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %2, %2, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %3, %3, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %4, %4, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %5, %5, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %6, %6, %8, %8 bitop3:0xde\n\tv_bitop3_b32 %7, %7, %8, %8 bitop3:0xde\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k10(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_alignbit_b32 %0, %0, %8, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_alignbit_b32 %4, %4, %8, 31\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_alignbit_b32 %0, %0, %8, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_alignbit_b32 %4, %4, %8, 31\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_alignbit_b32 %0, %0, %8, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_alignbit_b32 %4, %4, %8, 31\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_alignbit_b32 %7, %7, %8, 31\n\tv_alignbit_b32 %0, %0, %8, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_alignbit_b32 %4, %4, %8, 31\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_alignbit_b32 %7, %7, %8, 31\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k11(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_alignbit_b32 %0, %0, %8, %8\n\tv_alignbit_b32 %1, %1, %8, %8\n\tv_alignbit_b32 %2, %2, %8, %8\n\tv_alignbit_b32 %3, %3, %8, %8\n\tv_alignbit_b32 %4, %4, %8, %8\n\tv_alignbit_b32 %5, %5, %8, %8\n\tv_alignbit_b32 %6, %6, %8, %8\n\tv_alignbit_b32 %7, %7, %8, %8\n\tv_alignbit_b32 %0, %0, %8, %8\n\tv_alignbit_b32 %1, %1, %8, %8\n\tv_alignbit_b32 %2, %2, %8, %8\n\tv_alignbit_b32 %3, %3, %8, %8\n\tv_alignbit_b32 %4, %4, %8, %8\n\tv_alignbit_b32 %5, %5, %8, %8\n\tv_alignbit_b32 %6, %6, %8, %8\n\tv_alignbit_b32 %7, %7, %8, %8\n\tv_alignbit_b32 %0, %0, %8, %8\n\tv_alignbit_b32 %1, %1, %8, %8\n\tv_alignbit_b32 %2, %2, %8, %8\n\tv_alignbit_b32 %3, %3, %8, %8\n\tv_alignbit_b32 %4, %4, %8, %8\n\tv_alignbit_b32 %5, %5, %8, %8\n\tv_alignbit_b32 %6, %6, %8, %8\n\tv_alignbit_b32 %7, %7, %8, %8\n\tv_alignbit_b32 %0, %0, %8, %8\n\tv_alignbit_b32 %1, %1, %8, %8\n\tv_alignbit_b32 %2, %2, %8, %8\n\tv_alignbit_b32 %3, %3, %8, %8\n\tv_alignbit_b32 %4, %4, %8, %8\n\tv_alignbit_b32 %5, %5, %8, %8\n\tv_alignbit_b32 %6, %6, %8, %8\n\tv_alignbit_b32 %7, %7, %8, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k12(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_lshl_add_u32 %0, %0, 10, %8\n\tv_lshl_add_u32 %1, %1, 10, %8\n\tv_lshl_add_u32 %2, %2, 10, %8\n\tv_lshl_add_u32 %3, %3, 10, %8\n\tv_lshl_add_u32 %4, %4, 10, %8\n\tv_lshl_add_u32 %5, %5, 10, %8\n\tv_lshl_add_u32 %6, %6, 10, %8\n\tv_lshl_add_u32 %7, %7, 10, %8\n\tv_lshl_add_u32 %0, %0, 10, %8\n\tv_lshl_add_u32 %1, %1, 10, %8\n\tv_lshl_add_u32 %2, %2, 10, %8\n\tv_lshl_add_u32 %3, %3, 10, %8\n\tv_lshl_add_u32 %4, %4, 10, %8\n\tv_lshl_add_u32 %5, %5, 10, %8\n\tv_lshl_add_u32 %6, %6, 10, %8\n\tv_lshl_add_u32 %7, %7, 10, %8\n\tv_lshl_add_u32 %0, %0, 10, %8\n\tv_lshl_add_u32 %1, %1, 10, %8\n\tv_lshl_add_u32 %2, %2, 10, %8\n\tv_lshl_add_u32 %3, %3, 10, %8\n\tv_lshl_add_u32 %4, %4, 10, %8\n\tv_lshl_add_u32 %5, %5, 10, %8\n\tv_lshl_add_u32 %6, %6, 10, %8\n\tv_lshl_add_u32 %7, %7, 10, %8\n\tv_lshl_add_u32 %0, %0, 10, %8\n\tv_lshl_add_u32 %1, %1, 10, %8\n\tv_lshl_add_u32 %2, %2, 10, %8\n\tv_lshl_add_u32 %3, %3, 10, %8\n\tv_lshl_add_u32 %4, %4, 10, %8\n\tv_lshl_add_u32 %5, %5, 10, %8\n\tv_lshl_add_u32 %6, %6, 10, %8\n\tv_lshl_add_u32 %7, %7, 10, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k13(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_add3_u32 %0, %0, %8, %8\n\tv_add3_u32 %1, %1, %8, %8\n\tv_add3_u32 %2, %2, %8, %8\n\tv_add3_u32 %3, %3, %8, %8\n\tv_add3_u32 %4, %4, %8, %8\n\tv_add3_u32 %5, %5, %8, %8\n\tv_add3_u32 %6, %6, %8, %8\n\tv_add3_u32 %7, %7, %8, %8\n\tv_add3_u32 %0, %0, %8, %8\n\tv_add3_u32 %1, %1, %8, %8\n\tv_add3_u32 %2, %2, %8, %8\n\tv_add3_u32 %3, %3, %8, %8\n\tv_add3_u32 %4, %4, %8, %8\n\tv_add3_u32 %5, %5, %8, %8\n\tv_add3_u32 %6, %6, %8, %8\n\tv_add3_u32 %7, %7, %8, %8\n\tv_add3_u32 %0, %0, %8, %8\n\tv_add3_u32 %1, %1, %8, %8\n\tv_add3_u32 %2, %2, %8, %8\n\tv_add3_u32 %3, %3, %8, %8\n\tv_add3_u32 %4, %4, %8, %8\n\tv_add3_u32 %5, %5, %8, %8\n\tv_add3_u32 %6, %6, %8, %8\n\tv_add3_u32 %7, %7, %8, %8\n\tv_add3_u32 %0, %0, %8, %8\n\tv_add3_u32 %1, %1, %8, %8\n\tv_add3_u32 %2, %2, %8, %8\n\tv_add3_u32 %3, %3, %8, %8\n\tv_add3_u32 %4, %4, %8, %8\n\tv_add3_u32 %5, %5, %8, %8\n\tv_add3_u32 %6, %6, %8, %8\n\tv_add3_u32 %7, %7, %8, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k14(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_lshl_or_b32 %0, %0, 1, %8\n\tv_lshl_or_b32 %1, %1, 1, %8\n\tv_lshl_or_b32 %2, %2, 1, %8\n\tv_lshl_or_b32 %3, %3, 1, %8\n\tv_lshl_or_b32 %4, %4, 1, %8\n\tv_lshl_or_b32 %5, %5, 1, %8\n\tv_lshl_or_b32 %6, %6, 1, %8\n\tv_lshl_or_b32 %7, %7, 1, %8\n\tv_lshl_or_b32 %0, %0, 1, %8\n\tv_lshl_or_b32 %1, %1, 1, %8\n\tv_lshl_or_b32 %2, %2, 1, %8\n\tv_lshl_or_b32 %3, %3, 1, %8\n\tv_lshl_or_b32 %4, %4, 1, %8\n\tv_lshl_or_b32 %5, %5, 1, %8\n\tv_lshl_or_b32 %6, %6, 1, %8\n\tv_lshl_or_b32 %7, %7, 1, %8\n\tv_lshl_or_b32 %0, %0, 1, %8\n\tv_lshl_or_b32 %1, %1, 1, %8\n\tv_lshl_or_b32 %2, %2, 1, %8\n\tv_lshl_or_b32 %3, %3, 1, %8\n\tv_lshl_or_b32 %4, %4, 1, %8\n\tv_lshl_or_b32 %5, %5, 1, %8\n\tv_lshl_or_b32 %6, %6, 1, %8\n\tv_lshl_or_b32 %7, %7, 1, %8\n\tv_lshl_or_b32 %0, %0, 1, %8\n\tv_lshl_or_b32 %1, %1, 1, %8\n\tv_lshl_or_b32 %2, %2, 1, %8\n\tv_lshl_or_b32 %3, %3, 1, %8\n\tv_lshl_or_b32 %4, %4, 1, %8\n\tv_lshl_or_b32 %5, %5, 1, %8\n\tv_lshl_or_b32 %6, %6, 1, %8\n\tv_lshl_or_b32 %7, %7, 1, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k15(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_and_or_b32 %0, %0, %8, %8\n\tv_and_or_b32 %1, %1, %8, %8\n\tv_and_or_b32 %2, %2, %8, %8\n\tv_and_or_b32 %3, %3, %8, %8\n\tv_and_or_b32 %4, %4, %8, %8\n\tv_and_or_b32 %5, %5, %8, %8\n\tv_and_or_b32 %6, %6, %8, %8\n\tv_and_or_b32 %7, %7, %8, %8\n\tv_and_or_b32 %0, %0, %8, %8\n\tv_and_or_b32 %1, %1, %8, %8\n\tv_and_or_b32 %2, %2, %8, %8\n\tv_and_or_b32 %3, %3, %8, %8\n\tv_and_or_b32 %4, %4, %8, %8\n\tv_and_or_b32 %5, %5, %8, %8\n\tv_and_or_b32 %6, %6, %8, %8\n\tv_and_or_b32 %7, %7, %8, %8\n\tv_and_or_b32 %0, %0, %8, %8\n\tv_and_or_b32 %1, %1, %8, %8\n\tv_and_or_b32 %2, %2, %8, %8\n\tv_and_or_b32 %3, %3, %8, %8\n\tv_and_or_b32 %4, %4, %8, %8\n\tv_and_or_b32 %5, %5, %8, %8\n\tv_and_or_b32 %6, %6, %8, %8\n\tv_and_or_b32 %7, %7, %8, %8\n\tv_and_or_b32 %0, %0, %8, %8\n\tv_and_or_b32 %1, %1, %8, %8\n\tv_and_or_b32 %2, %2, %8, %8\n\tv_and_or_b32 %3, %3, %8, %8\n\tv_and_or_b32 %4, %4, %8, %8\n\tv_and_or_b32 %5, %5, %8, %8\n\tv_and_or_b32 %6, %6, %8, %8\n\tv_and_or_b32 %7, %7, %8, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k16(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_bfe_u32 %0, %0, 3, 1\n\tv_bfe_u32 %1, %1, 3, 1\n\tv_bfe_u32 %2, %2, 3, 1\n\tv_bfe_u32 %3, %3, 3, 1\n\tv_bfe_u32 %4, %4, 3, 1\n\tv_bfe_u32 %5, %5, 3, 1\n\tv_bfe_u32 %6, %6, 3, 1\n\tv_bfe_u32 %7, %7, 3, 1\n\tv_bfe_u32 %0, %0, 3, 1\n\tv_bfe_u32 %1, %1, 3, 1\n\tv_bfe_u32 %2, %2, 3, 1\n\tv_bfe_u32 %3, %3, 3, 1\n\tv_bfe_u32 %4, %4, 3, 1\n\tv_bfe_u32 %5, %5, 3, 1\n\tv_bfe_u32 %6, %6, 3, 1\n\tv_bfe_u32 %7, %7, 3, 1\n\tv_bfe_u32 %0, %0, 3, 1\n\tv_bfe_u32 %1, %1, 3, 1\n\tv_bfe_u32 %2, %2, 3, 1\n\tv_bfe_u32 %3, %3, 3, 1\n\tv_bfe_u32 %4, %4, 3, 1\n\tv_bfe_u32 %5, %5, 3, 1\n\tv_bfe_u32 %6, %6, 3, 1\n\tv_bfe_u32 %7, %7, 3, 1\n\tv_bfe_u32 %0, %0, 3, 1\n\tv_bfe_u32 %1, %1, 3, 1\n\tv_bfe_u32 %2, %2, 3, 1\n\tv_bfe_u32 %3, %3, 3, 1\n\tv_bfe_u32 %4, %4, 3, 1\n\tv_bfe_u32 %5, %5, 3, 1\n\tv_bfe_u32 %6, %6, 3, 1\n\tv_bfe_u32 %7, %7, 3, 1\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k17(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_bfe_u32 %0, %0, %8, 1\n\tv_bfe_u32 %1, %1, %8, 1\n\tv_bfe_u32 %2, %2, %8, 1\n\tv_bfe_u32 %3, %3, %8, 1\n\tv_bfe_u32 %4, %4, %8, 1\n\tv_bfe_u32 %5, %5, %8, 1\n\tv_bfe_u32 %6, %6, %8, 1\n\tv_bfe_u32 %7, %7, %8, 1\n\tv_bfe_u32 %0, %0, %8, 1\n\tv_bfe_u32 %1, %1, %8, 1\n\tv_bfe_u32 %2, %2, %8, 1\n\tv_bfe_u32 %3, %3, %8, 1\n\tv_bfe_u32 %4, %4, %8, 1\n\tv_bfe_u32 %5, %5, %8, 1\n\tv_bfe_u32 %6, %6, %8, 1\n\tv_bfe_u32 %7, %7, %8, 1\n\tv_bfe_u32 %0, %0, %8, 1\n\tv_bfe_u32 %1, %1, %8, 1\n\tv_bfe_u32 %2, %2, %8, 1\n\tv_bfe_u32 %3, %3, %8, 1\n\tv_bfe_u32 %4, %4, %8, 1\n\tv_bfe_u32 %5, %5, %8, 1\n\tv_bfe_u32 %6, %6, %8, 1\n\tv_bfe_u32 %7, %7, %8, 1\n\tv_bfe_u32 %0, %0, %8, 1\n\tv_bfe_u32 %1, %1, %8, 1\n\tv_bfe_u32 %2, %2, %8, 1\n\tv_bfe_u32 %3, %3, %8, 1\n\tv_bfe_u32 %4, %4, %8, 1\n\tv_bfe_u32 %5, %5, %8, 1\n\tv_bfe_u32 %6, %6, %8, 1\n\tv_bfe_u32 %7, %7, %8, 1\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k18(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_mad_u32_u24 %0, %0, %8, %8\n\tv_mad_u32_u24 %1, %1, %8, %8\n\tv_mad_u32_u24 %2, %2, %8, %8\n\tv_mad_u32_u24 %3, %3, %8, %8\n\tv_mad_u32_u24 %4, %4, %8, %8\n\tv_mad_u32_u24 %5, %5, %8, %8\n\tv_mad_u32_u24 %6, %6, %8, %8\n\tv_mad_u32_u24 %7, %7, %8, %8\n\tv_mad_u32_u24 %0, %0, %8, %8\n\tv_mad_u32_u24 %1, %1, %8, %8\n\tv_mad_u32_u24 %2, %2, %8, %8\n\tv_mad_u32_u24 %3, %3, %8, %8\n\tv_mad_u32_u24 %4, %4, %8, %8\n\tv_mad_u32_u24 %5, %5, %8, %8\n\tv_mad_u32_u24 %6, %6, %8, %8\n\tv_mad_u32_u24 %7, %7, %8, %8\n\tv_mad_u32_u24 %0, %0, %8, %8\n\tv_mad_u32_u24 %1, %1, %8, %8\n\tv_mad_u32_u24 %2, %2, %8, %8\n\tv_mad_u32_u24 %3, %3, %8, %8\n\tv_mad_u32_u24 %4, %4, %8, %8\n\tv_mad_u32_u24 %5, %5, %8, %8\n\tv_mad_u32_u24 %6, %6, %8, %8\n\tv_mad_u32_u24 %7, %7, %8, %8\n\tv_mad_u32_u24 %0, %0, %8, %8\n\tv_mad_u32_u24 %1, %1, %8, %8\n\tv_mad_u32_u24 %2, %2, %8, %8\n\tv_mad_u32_u24 %3, %3, %8, %8\n\tv_mad_u32_u24 %4, %4, %8, %8\n\tv_mad_u32_u24 %5, %5, %8, %8\n\tv_mad_u32_u24 %6, %6, %8, %8\n\tv_mad_u32_u24 %7, %7, %8, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k19(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_mul_lo_u32 %0, %0, %8\n\tv_mul_lo_u32 %1, %1, %8\n\tv_mul_lo_u32 %2, %2, %8\n\tv_mul_lo_u32 %3, %3, %8\n\tv_mul_lo_u32 %4, %4, %8\n\tv_mul_lo_u32 %5, %5, %8\n\tv_mul_lo_u32 %6, %6, %8\n\tv_mul_lo_u32 %7, %7, %8\n\tv_mul_lo_u32 %0, %0, %8\n\tv_mul_lo_u32 %1, %1, %8\n\tv_mul_lo_u32 %2, %2, %8\n\tv_mul_lo_u32 %3, %3, %8\n\tv_mul_lo_u32 %4, %4, %8\n\tv_mul_lo_u32 %5, %5, %8\n\tv_mul_lo_u32 %6, %6, %8\n\tv_mul_lo_u32 %7, %7, %8\n\tv_mul_lo_u32 %0, %0, %8\n\tv_mul_lo_u32 %1, %1, %8\n\tv_mul_lo_u32 %2, %2, %8\n\tv_mul_lo_u32 %3, %3, %8\n\tv_mul_lo_u32 %4, %4, %8\n\tv_mul_lo_u32 %5, %5, %8\n\tv_mul_lo_u32 %6, %6, %8\n\tv_mul_lo_u32 %7, %7, %8\n\tv_mul_lo_u32 %0, %0, %8\n\tv_mul_lo_u32 %1, %1, %8\n\tv_mul_lo_u32 %2, %2, %8\n\tv_mul_lo_u32 %3, %3, %8\n\tv_mul_lo_u32 %4, %4, %8\n\tv_mul_lo_u32 %5, %5, %8\n\tv_mul_lo_u32 %6, %6, %8\n\tv_mul_lo_u32 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k20(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_bcnt_u32_b32 %0, %0, %8\n\tv_bcnt_u32_b32 %1, %1, %8\n\tv_bcnt_u32_b32 %2, %2, %8\n\tv_bcnt_u32_b32 %3, %3, %8\n\tv_bcnt_u32_b32 %4, %4, %8\n\tv_bcnt_u32_b32 %5, %5, %8\n\tv_bcnt_u32_b32 %6, %6, %8\n\tv_bcnt_u32_b32 %7, %7, %8\n\tv_bcnt_u32_b32 %0, %0, %8\n\tv_bcnt_u32_b32 %1, %1, %8\n\tv_bcnt_u32_b32 %2, %2, %8\n\tv_bcnt_u32_b32 %3, %3, %8\n\tv_bcnt_u32_b32 %4, %4, %8\n\tv_bcnt_u32_b32 %5, %5, %8\n\tv_bcnt_u32_b32 %6, %6, %8\n\tv_bcnt_u32_b32 %7, %7, %8\n\tv_bcnt_u32_b32 %0, %0, %8\n\tv_bcnt_u32_b32 %1, %1, %8\n\tv_bcnt_u32_b32 %2, %2, %8\n\tv_bcnt_u32_b32 %3, %3, %8\n\tv_bcnt_u32_b32 %4, %4, %8\n\tv_bcnt_u32_b32 %5, %5, %8\n\tv_bcnt_u32_b32 %6, %6, %8\n\tv_bcnt_u32_b32 %7, %7, %8\n\tv_bcnt_u32_b32 %0, %0, %8\n\tv_bcnt_u32_b32 %1, %1, %8\n\tv_bcnt_u32_b32 %2, %2, %8\n\tv_bcnt_u32_b32 %3, %3, %8\n\tv_bcnt_u32_b32 %4, %4, %8\n\tv_bcnt_u32_b32 %5, %5, %8\n\tv_bcnt_u32_b32 %6, %6, %8\n\tv_bcnt_u32_b32 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k21(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_add_co_u32 %0, vcc, %0, %0\n\tv_add_co_u32 %1, vcc, %1, %1\n\tv_add_co_u32 %2, vcc, %2, %2\n\tv_add_co_u32 %3, vcc, %3, %3\n\tv_add_co_u32 %4, vcc, %4, %4\n\tv_add_co_u32 %5, vcc, %5, %5\n\tv_add_co_u32 %6, vcc, %6, %6\n\tv_add_co_u32 %7, vcc, %7, %7\n\tv_add_co_u32 %0, vcc, %0, %0\n\tv_add_co_u32 %1, vcc, %1, %1\n\tv_add_co_u32 %2, vcc, %2, %2\n\tv_add_co_u32 %3, vcc, %3, %3\n\tv_add_co_u32 %4, vcc, %4, %4\n\tv_add_co_u32 %5, vcc, %5, %5\n\tv_add_co_u32 %6, vcc, %6, %6\n\tv_add_co_u32 %7, vcc, %7, %7\n\tv_add_co_u32 %0, vcc, %0, %0\n\tv_add_co_u32 %1, vcc, %1, %1\n\tv_add_co_u32 %2, vcc, %2, %2\n\tv_add_co_u32 %3, vcc, %3, %3\n\tv_add_co_u32 %4, vcc, %4, %4\n\tv_add_co_u32 %5, vcc, %5, %5\n\tv_add_co_u32 %6, vcc, %6, %6\n\tv_add_co_u32 %7, vcc, %7, %7\n\tv_add_co_u32 %0, vcc, %0, %0\n\tv_add_co_u32 %1, vcc, %1, %1\n\tv_add_co_u32 %2, vcc, %2, %2\n\tv_add_co_u32 %3, vcc, %3, %3\n\tv_add_co_u32 %4, vcc, %4, %4\n\tv_add_co_u32 %5, vcc, %5, %5\n\tv_add_co_u32 %6, vcc, %6, %6\n\tv_add_co_u32 %7, vcc, %7, %7\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k22(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc\n\tv_addc_co_u32 %4, vcc, 0, %4, vcc\n\tv_addc_co_u32 %5, vcc, 0, %5, vcc\n\tv_addc_co_u32 %6, vcc, 0, %6, vcc\n\tv_addc_co_u32 %7, vcc, 0, %7, vcc\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc\n\tv_addc_co_u32 %4, vcc, 0, %4, vcc\n\tv_addc_co_u32 %5, vcc, 0, %5, vcc\n\tv_addc_co_u32 %6, vcc, 0, %6, vcc\n\tv_addc_co_u32 %7, vcc, 0, %7, vcc\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc\n\tv_addc_co_u32 %4, vcc, 0, %4, vcc\n\tv_addc_co_u32 %5, vcc, 0, %5, vcc\n\tv_addc_co_u32 %6, vcc, 0, %6, vcc\n\tv_addc_co_u32 %7, vcc, 0, %7, vcc\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc\n\tv_addc_co_u32 %4, vcc, 0, %4, vcc\n\tv_addc_co_u32 %5, vcc, 0, %5, vcc\n\tv_addc_co_u32 %6, vcc, 0, %6, vcc\n\tv_addc_co_u32 %7, vcc, 0, %7, vcc\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k23(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\tv_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc\n\tv_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\tv_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc\n\tv_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\tv_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc\n\tv_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\tv_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k24(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_cmp_lt_u32 vcc, %0, %8\n\tv_cmp_lt_u32 vcc, %1, %8\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cmp_lt_u32 vcc, %3, %8\n\tv_cmp_lt_u32 vcc, %4, %8\n\tv_cmp_lt_u32 vcc, %5, %8\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cmp_lt_u32 vcc, %7, %8\n\tv_cmp_lt_u32 vcc, %0, %8\n\tv_cmp_lt_u32 vcc, %1, %8\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cmp_lt_u32 vcc, %3, %8\n\tv_cmp_lt_u32 vcc, %4, %8\n\tv_cmp_lt_u32 vcc, %5, %8\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cmp_lt_u32 vcc, %7, %8\n\tv_cmp_lt_u32 vcc, %0, %8\n\tv_cmp_lt_u32 vcc, %1, %8\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cmp_lt_u32 vcc, %3, %8\n\tv_cmp_lt_u32 vcc, %4, %8\n\tv_cmp_lt_u32 vcc, %5, %8\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cmp_lt_u32 vcc, %7, %8\n\tv_cmp_lt_u32 vcc, %0, %8\n\tv_cmp_lt_u32 vcc, %1, %8\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cmp_lt_u32 vcc, %3, %8\n\tv_cmp_lt_u32 vcc, %4, %8\n\tv_cmp_lt_u32 vcc, %5, %8\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cmp_lt_u32 vcc, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k25(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_sub_u32 %0, %0, %8\n\tv_sub_u32 %1, %1, %8\n\tv_sub_u32 %2, %2, %8\n\tv_sub_u32 %3, %3, %8\n\tv_sub_u32 %4, %4, %8\n\tv_sub_u32 %5, %5, %8\n\tv_sub_u32 %6, %6, %8\n\tv_sub_u32 %7, %7, %8\n\tv_sub_u32 %0, %0, %8\n\tv_sub_u32 %1, %1, %8\n\tv_sub_u32 %2, %2, %8\n\tv_sub_u32 %3, %3, %8\n\tv_sub_u32 %4, %4, %8\n\tv_sub_u32 %5, %5, %8\n\tv_sub_u32 %6, %6, %8\n\tv_sub_u32 %7, %7, %8\n\tv_sub_u32 %0, %0, %8\n\tv_sub_u32 %1, %1, %8\n\tv_sub_u32 %2, %2, %8\n\tv_sub_u32 %3, %3, %8\n\tv_sub_u32 %4, %4, %8\n\tv_sub_u32 %5, %5, %8\n\tv_sub_u32 %6, %6, %8\n\tv_sub_u32 %7, %7, %8\n\tv_sub_u32 %0, %0, %8\n\tv_sub_u32 %1, %1, %8\n\tv_sub_u32 %2, %2, %8\n\tv_sub_u32 %3, %3, %8\n\tv_sub_u32 %4, %4, %8\n\tv_sub_u32 %5, %5, %8\n\tv_sub_u32 %6, %6, %8\n\tv_sub_u32 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k26(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_xad_u32 %0, %0, %8, %8\n\tv_xad_u32 %1, %1, %8, %8\n\tv_xad_u32 %2, %2, %8, %8\n\tv_xad_u32 %3, %3, %8, %8\n\tv_xad_u32 %4, %4, %8, %8\n\tv_xad_u32 %5, %5, %8, %8\n\tv_xad_u32 %6, %6, %8, %8\n\tv_xad_u32 %7, %7, %8, %8\n\tv_xad_u32 %0, %0, %8, %8\n\tv_xad_u32 %1, %1, %8, %8\n\tv_xad_u32 %2, %2, %8, %8\n\tv_xad_u32 %3, %3, %8, %8\n\tv_xad_u32 %4, %4, %8, %8\n\tv_xad_u32 %5, %5, %8, %8\n\tv_xad_u32 %6, %6, %8, %8\n\tv_xad_u32 %7, %7, %8, %8\n\tv_xad_u32 %0, %0, %8, %8\n\tv_xad_u32 %1, %1, %8, %8\n\tv_xad_u32 %2, %2, %8, %8\n\tv_xad_u32 %3, %3, %8, %8\n\tv_xad_u32 %4, %4, %8, %8\n\tv_xad_u32 %5, %5, %8, %8\n\tv_xad_u32 %6, %6, %8, %8\n\tv_xad_u32 %7, %7, %8, %8\n\tv_xad_u32 %0, %0, %8, %8\n\tv_xad_u32 %1, %1, %8, %8\n\tv_xad_u32 %2, %2, %8, %8\n\tv_xad_u32 %3, %3, %8, %8\n\tv_xad_u32 %4, %4, %8, %8\n\tv_xad_u32 %5, %5, %8, %8\n\tv_xad_u32 %6, %6, %8, %8\n\tv_xad_u32 %7, %7, %8, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k27(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_perm_b32 %0, %0, %8, %8\n\tv_perm_b32 %1, %1, %8, %8\n\tv_perm_b32 %2, %2, %8, %8\n\tv_perm_b32 %3, %3, %8, %8\n\tv_perm_b32 %4, %4, %8, %8\n\tv_perm_b32 %5, %5, %8, %8\n\tv_perm_b32 %6, %6, %8, %8\n\tv_perm_b32 %7, %7, %8, %8\n\tv_perm_b32 %0, %0, %8, %8\n\tv_perm_b32 %1, %1, %8, %8\n\tv_perm_b32 %2, %2, %8, %8\n\tv_perm_b32 %3, %3, %8, %8\n\tv_perm_b32 %4, %4, %8, %8\n\tv_perm_b32 %5, %5, %8, %8\n\tv_perm_b32 %6, %6, %8, %8\n\tv_perm_b32 %7, %7, %8, %8\n\tv_perm_b32 %0, %0, %8, %8\n\tv_perm_b32 %1, %1, %8, %8\n\tv_perm_b32 %2, %2, %8, %8\n\tv_perm_b32 %3, %3, %8, %8\n\tv_perm_b32 %4, %4, %8, %8\n\tv_perm_b32 %5, %5, %8, %8\n\tv_perm_b32 %6, %6, %8, %8\n\tv_perm_b32 %7, %7, %8, %8\n\tv_perm_b32 %0, %0, %8, %8\n\tv_perm_b32 %1, %1, %8, %8\n\tv_perm_b32 %2, %2, %8, %8\n\tv_perm_b32 %3, %3, %8, %8\n\tv_perm_b32 %4, %4, %8, %8\n\tv_perm_b32 %5, %5, %8, %8\n\tv_perm_b32 %6, %6, %8, %8\n\tv_perm_b32 %7, %7, %8, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k28(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_mov_b32 %0, %8\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %8\n\tv_mov_b32 %0, %8\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %8\n\tv_mov_b32 %0, %8\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %8\n\tv_mov_b32 %0, %8\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k29(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_alignbyte_b32 %0, %0, %8, 1\n\tv_alignbyte_b32 %1, %1, %8, 1\n\tv_alignbyte_b32 %2, %2, %8, 1\n\tv_alignbyte_b32 %3, %3, %8, 1\n\tv_alignbyte_b32 %4, %4, %8, 1\n\tv_alignbyte_b32 %5, %5, %8, 1\n\tv_alignbyte_b32 %6, %6, %8, 1\n\tv_alignbyte_b32 %7, %7, %8, 1\n\tv_alignbyte_b32 %0, %0, %8, 1\n\tv_alignbyte_b32 %1, %1, %8, 1\n\tv_alignbyte_b32 %2, %2, %8, 1\n\tv_alignbyte_b32 %3, %3, %8, 1\n\tv_alignbyte_b32 %4, %4, %8, 1\n\tv_alignbyte_b32 %5, %5, %8, 1\n\tv_alignbyte_b32 %6, %6, %8, 1\n\tv_alignbyte_b32 %7, %7, %8, 1\n\tv_alignbyte_b32 %0, %0, %8, 1\n\tv_alignbyte_b32 %1, %1, %8, 1\n\tv_alignbyte_b32 %2, %2, %8, 1\n\tv_alignbyte_b32 %3, %3, %8, 1\n\tv_alignbyte_b32 %4, %4, %8, 1\n\tv_alignbyte_b32 %5, %5, %8, 1\n\tv_alignbyte_b32 %6, %6, %8, 1\n\tv_alignbyte_b32 %7, %7, %8, 1\n\tv_alignbyte_b32 %0, %0, %8, 1\n\tv_alignbyte_b32 %1, %1, %8, 1\n\tv_alignbyte_b32 %2, %2, %8, 1\n\tv_alignbyte_b32 %3, %3, %8, 1\n\tv_alignbyte_b32 %4, %4, %8, 1\n\tv_alignbyte_b32 %5, %5, %8, 1\n\tv_alignbyte_b32 %6, %6, %8, 1\n\tv_alignbyte_b32 %7, %7, %8, 1\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k30(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_pk_add_u16 %0, %0, %8\n\tv_pk_add_u16 %1, %1, %8\n\tv_pk_add_u16 %2, %2, %8\n\tv_pk_add_u16 %3, %3, %8\n\tv_pk_add_u16 %4, %4, %8\n\tv_pk_add_u16 %5, %5, %8\n\tv_pk_add_u16 %6, %6, %8\n\tv_pk_add_u16 %7, %7, %8\n\tv_pk_add_u16 %0, %0, %8\n\tv_pk_add_u16 %1, %1, %8\n\tv_pk_add_u16 %2, %2, %8\n\tv_pk_add_u16 %3, %3, %8\n\tv_pk_add_u16 %4, %4, %8\n\tv_pk_add_u16 %5, %5, %8\n\tv_pk_add_u16 %6, %6, %8\n\tv_pk_add_u16 %7, %7, %8\n\tv_pk_add_u16 %0, %0, %8\n\tv_pk_add_u16 %1, %1, %8\n\tv_pk_add_u16 %2, %2, %8\n\tv_pk_add_u16 %3, %3, %8\n\tv_pk_add_u16 %4, %4, %8\n\tv_pk_add_u16 %5, %5, %8\n\tv_pk_add_u16 %6, %6, %8\n\tv_pk_add_u16 %7, %7, %8\n\tv_pk_add_u16 %0, %0, %8\n\tv_pk_add_u16 %1, %1, %8\n\tv_pk_add_u16 %2, %2, %8\n\tv_pk_add_u16 %3, %3, %8\n\tv_pk_add_u16 %4, %4, %8\n\tv_pk_add_u16 %5, %5, %8\n\tv_pk_add_u16 %6, %6, %8\n\tv_pk_add_u16 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k31(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_ffbh_u32 %0, %0\n\tv_ffbh_u32 %1, %1\n\tv_ffbh_u32 %2, %2\n\tv_ffbh_u32 %3, %3\n\tv_ffbh_u32 %4, %4\n\tv_ffbh_u32 %5, %5\n\tv_ffbh_u32 %6, %6\n\tv_ffbh_u32 %7, %7\n\tv_ffbh_u32 %0, %0\n\tv_ffbh_u32 %1, %1\n\tv_ffbh_u32 %2, %2\n\tv_ffbh_u32 %3, %3\n\tv_ffbh_u32 %4, %4\n\tv_ffbh_u32 %5, %5\n\tv_ffbh_u32 %6, %6\n\tv_ffbh_u32 %7, %7\n\tv_ffbh_u32 %0, %0\n\tv_ffbh_u32 %1, %1\n\tv_ffbh_u32 %2, %2\n\tv_ffbh_u32 %3, %3\n\tv_ffbh_u32 %4, %4\n\tv_ffbh_u32 %5, %5\n\tv_ffbh_u32 %6, %6\n\tv_ffbh_u32 %7, %7\n\tv_ffbh_u32 %0, %0\n\tv_ffbh_u32 %1, %1\n\tv_ffbh_u32 %2, %2\n\tv_ffbh_u32 %3, %3\n\tv_ffbh_u32 %4, %4\n\tv_ffbh_u32 %5, %5\n\tv_ffbh_u32 %6, %6\n\tv_ffbh_u32 %7, %7\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k32(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_bfrev_b32 %0, %0\n\tv_bfrev_b32 %1, %1\n\tv_bfrev_b32 %2, %2\n\tv_bfrev_b32 %3, %3\n\tv_bfrev_b32 %4, %4\n\tv_bfrev_b32 %5, %5\n\tv_bfrev_b32 %6, %6\n\tv_bfrev_b32 %7, %7\n\tv_bfrev_b32 %0, %0\n\tv_bfrev_b32 %1, %1\n\tv_bfrev_b32 %2, %2\n\tv_bfrev_b32 %3, %3\n\tv_bfrev_b32 %4, %4\n\tv_bfrev_b32 %5, %5\n\tv_bfrev_b32 %6, %6\n\tv_bfrev_b32 %7, %7\n\tv_bfrev_b32 %0, %0\n\tv_bfrev_b32 %1, %1\n\tv_bfrev_b32 %2, %2\n\tv_bfrev_b32 %3, %3\n\tv_bfrev_b32 %4, %4\n\tv_bfrev_b32 %5, %5\n\tv_bfrev_b32 %6, %6\n\tv_bfrev_b32 %7, %7\n\tv_bfrev_b32 %0, %0\n\tv_bfrev_b32 %1, %1\n\tv_bfrev_b32 %2, %2\n\tv_bfrev_b32 %3, %3\n\tv_bfrev_b32 %4, %4\n\tv_bfrev_b32 %5, %5\n\tv_bfrev_b32 %6, %6\n\tv_bfrev_b32 %7, %7\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k33(uint32_t *out, int iters, uint32_t seed) {
    uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %1, 1, %1\n\tv_lshlrev_b64 %2, 1, %2\n\tv_lshlrev_b64 %3, 1, %3\n\tv_lshlrev_b64 %4, 1, %4\n\tv_lshlrev_b64 %5, 1, %5\n\tv_lshlrev_b64 %6, 1, %6\n\tv_lshlrev_b64 %7, 1, %7\n\tv_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %1, 1, %1\n\tv_lshlrev_b64 %2, 1, %2\n\tv_lshlrev_b64 %3, 1, %3\n\tv_lshlrev_b64 %4, 1, %4\n\tv_lshlrev_b64 %5, 1, %5\n\tv_lshlrev_b64 %6, 1, %6\n\tv_lshlrev_b64 %7, 1, %7\n\tv_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %1, 1, %1\n\tv_lshlrev_b64 %2, 1, %2\n\tv_lshlrev_b64 %3, 1, %3\n\tv_lshlrev_b64 %4, 1, %4\n\tv_lshlrev_b64 %5, 1, %5\n\tv_lshlrev_b64 %6, 1, %6\n\tv_lshlrev_b64 %7, 1, %7\n\tv_lshlrev_b64 %0, 1, %0\n\tv_lshlrev_b64 %1, 1, %1\n\tv_lshlrev_b64 %2, 1, %2\n\tv_lshlrev_b64 %3, 1, %3\n\tv_lshlrev_b64 %4, 1, %4\n\tv_lshlrev_b64 %5, 1, %5\n\tv_lshlrev_b64 %6, 1, %6\n\tv_lshlrev_b64 %7, 1, %7\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k34(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_ashrrev_i32 %0, 31, %0\n\tv_ashrrev_i32 %1, 31, %1\n\tv_ashrrev_i32 %2, 31, %2\n\tv_ashrrev_i32 %3, 31, %3\n\tv_ashrrev_i32 %4, 31, %4\n\tv_ashrrev_i32 %5, 31, %5\n\tv_ashrrev_i32 %6, 31, %6\n\tv_ashrrev_i32 %7, 31, %7\n\tv_ashrrev_i32 %0, 31, %0\n\tv_ashrrev_i32 %1, 31, %1\n\tv_ashrrev_i32 %2, 31, %2\n\tv_ashrrev_i32 %3, 31, %3\n\tv_ashrrev_i32 %4, 31, %4\n\tv_ashrrev_i32 %5, 31, %5\n\tv_ashrrev_i32 %6, 31, %6\n\tv_ashrrev_i32 %7, 31, %7\n\tv_ashrrev_i32 %0, 31, %0\n\tv_ashrrev_i32 %1, 31, %1\n\tv_ashrrev_i32 %2, 31, %2\n\tv_ashrrev_i32 %3, 31, %3\n\tv_ashrrev_i32 %4, 31, %4\n\tv_ashrrev_i32 %5, 31, %5\n\tv_ashrrev_i32 %6, 31, %6\n\tv_ashrrev_i32 %7, 31, %7\n\tv_ashrrev_i32 %0, 31, %0\n\tv_ashrrev_i32 %1, 31, %1\n\tv_ashrrev_i32 %2, 31, %2\n\tv_ashrrev_i32 %3, 31, %3\n\tv_ashrrev_i32 %4, 31, %4\n\tv_ashrrev_i32 %5, 31, %5\n\tv_ashrrev_i32 %6, 31, %6\n\tv_ashrrev_i32 %7, 31, %7\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k35(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_lshlrev_b32 %0, %8, %0\n\tv_lshlrev_b32 %1, %8, %1\n\tv_lshlrev_b32 %2, %8, %2\n\tv_lshlrev_b32 %3, %8, %3\n\tv_lshlrev_b32 %4, %8, %4\n\tv_lshlrev_b32 %5, %8, %5\n\tv_lshlrev_b32 %6, %8, %6\n\tv_lshlrev_b32 %7, %8, %7\n\tv_lshlrev_b32 %0, %8, %0\n\tv_lshlrev_b32 %1, %8, %1\n\tv_lshlrev_b32 %2, %8, %2\n\tv_lshlrev_b32 %3, %8, %3\n\tv_lshlrev_b32 %4, %8, %4\n\tv_lshlrev_b32 %5, %8, %5\n\tv_lshlrev_b32 %6, %8, %6\n\tv_lshlrev_b32 %7, %8, %7\n\tv_lshlrev_b32 %0, %8, %0\n\tv_lshlrev_b32 %1, %8, %1\n\tv_lshlrev_b32 %2, %8, %2\n\tv_lshlrev_b32 %3, %8, %3\n\tv_lshlrev_b32 %4, %8, %4\n\tv_lshlrev_b32 %5, %8, %5\n\tv_lshlrev_b32 %6, %8, %6\n\tv_lshlrev_b32 %7, %8, %7\n\tv_lshlrev_b32 %0, %8, %0\n\tv_lshlrev_b32 %1, %8, %1\n\tv_lshlrev_b32 %2, %8, %2\n\tv_lshlrev_b32 %3, %8, %3\n\tv_lshlrev_b32 %4, %8, %4\n\tv_lshlrev_b32 %5, %8, %5\n\tv_lshlrev_b32 %6, %8, %6\n\tv_lshlrev_b32 %7, %8, %7\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k36(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_lshrrev_b32 %0, %8, %0\n\tv_lshrrev_b32 %1, %8, %1\n\tv_lshrrev_b32 %2, %8, %2\n\tv_lshrrev_b32 %3, %8, %3\n\tv_lshrrev_b32 %4, %8, %4\n\tv_lshrrev_b32 %5, %8, %5\n\tv_lshrrev_b32 %6, %8, %6\n\tv_lshrrev_b32 %7, %8, %7\n\tv_lshrrev_b32 %0, %8, %0\n\tv_lshrrev_b32 %1, %8, %1\n\tv_lshrrev_b32 %2, %8, %2\n\tv_lshrrev_b32 %3, %8, %3\n\tv_lshrrev_b32 %4, %8, %4\n\tv_lshrrev_b32 %5, %8, %5\n\tv_lshrrev_b32 %6, %8, %6\n\tv_lshrrev_b32 %7, %8, %7\n\tv_lshrrev_b32 %0, %8, %0\n\tv_lshrrev_b32 %1, %8, %1\n\tv_lshrrev_b32 %2, %8, %2\n\tv_lshrrev_b32 %3, %8, %3\n\tv_lshrrev_b32 %4, %8, %4\n\tv_lshrrev_b32 %5, %8, %5\n\tv_lshrrev_b32 %6, %8, %6\n\tv_lshrrev_b32 %7, %8, %7\n\tv_lshrrev_b32 %0, %8, %0\n\tv_lshrrev_b32 %1, %8, %1\n\tv_lshrrev_b32 %2, %8, %2\n\tv_lshrrev_b32 %3, %8, %3\n\tv_lshrrev_b32 %4, %8, %4\n\tv_lshrrev_b32 %5, %8, %5\n\tv_lshrrev_b32 %6, %8, %6\n\tv_lshrrev_b32 %7, %8, %7\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k37(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_not_b32 %0, %0\n\tv_not_b32 %1, %1\n\tv_not_b32 %2, %2\n\tv_not_b32 %3, %3\n\tv_not_b32 %4, %4\n\tv_not_b32 %5, %5\n\tv_not_b32 %6, %6\n\tv_not_b32 %7, %7\n\tv_not_b32 %0, %0\n\tv_not_b32 %1, %1\n\tv_not_b32 %2, %2\n\tv_not_b32 %3, %3\n\tv_not_b32 %4, %4\n\tv_not_b32 %5, %5\n\tv_not_b32 %6, %6\n\tv_not_b32 %7, %7\n\tv_not_b32 %0, %0\n\tv_not_b32 %1, %1\n\tv_not_b32 %2, %2\n\tv_not_b32 %3, %3\n\tv_not_b32 %4, %4\n\tv_not_b32 %5, %5\n\tv_not_b32 %6, %6\n\tv_not_b32 %7, %7\n\tv_not_b32 %0, %0\n\tv_not_b32 %1, %1\n\tv_not_b32 %2, %2\n\tv_not_b32 %3, %3\n\tv_not_b32 %4, %4\n\tv_not_b32 %5, %5\n\tv_not_b32 %6, %6\n\tv_not_b32 %7, %7\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k38(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_add_u32 %0, %0, %0\n\tv_add_u32 %1, %1, %1\n\tv_add_u32 %2, %2, %2\n\tv_add_u32 %3, %3, %3\n\tv_add_u32 %4, %4, %4\n\tv_add_u32 %5, %5, %5\n\tv_add_u32 %6, %6, %6\n\tv_add_u32 %7, %7, %7\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %1, %1, %1\n\tv_add_u32 %2, %2, %2\n\tv_add_u32 %3, %3, %3\n\tv_add_u32 %4, %4, %4\n\tv_add_u32 %5, %5, %5\n\tv_add_u32 %6, %6, %6\n\tv_add_u32 %7, %7, %7\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %1, %1, %1\n\tv_add_u32 %2, %2, %2\n\tv_add_u32 %3, %3, %3\n\tv_add_u32 %4, %4, %4\n\tv_add_u32 %5, %5, %5\n\tv_add_u32 %6, %6, %6\n\tv_add_u32 %7, %7, %7\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %1, %1, %1\n\tv_add_u32 %2, %2, %2\n\tv_add_u32 %3, %3, %3\n\tv_add_u32 %4, %4, %4\n\tv_add_u32 %5, %5, %5\n\tv_add_u32 %6, %6, %6\n\tv_add_u32 %7, %7, %7\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k39(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_add_u32 %0, 0x12345, %0\n\tv_add_u32 %1, 0x12345, %1\n\tv_add_u32 %2, 0x12345, %2\n\tv_add_u32 %3, 0x12345, %3\n\tv_add_u32 %4, 0x12345, %4\n\tv_add_u32 %5, 0x12345, %5\n\tv_add_u32 %6, 0x12345, %6\n\tv_add_u32 %7, 0x12345, %7\n\tv_add_u32 %0, 0x12345, %0\n\tv_add_u32 %1, 0x12345, %1\n\tv_add_u32 %2, 0x12345, %2\n\tv_add_u32 %3, 0x12345, %3\n\tv_add_u32 %4, 0x12345, %4\n\tv_add_u32 %5, 0x12345, %5\n\tv_add_u32 %6, 0x12345, %6\n\tv_add_u32 %7, 0x12345, %7\n\tv_add_u32 %0, 0x12345, %0\n\tv_add_u32 %1, 0x12345, %1\n\tv_add_u32 %2, 0x12345, %2\n\tv_add_u32 %3, 0x12345, %3\n\tv_add_u32 %4, 0x12345, %4\n\tv_add_u32 %5, 0x12345, %5\n\tv_add_u32 %6, 0x12345, %6\n\tv_add_u32 %7, 0x12345, %7\n\tv_add_u32 %0, 0x12345, %0\n\tv_add_u32 %1, 0x12345, %1\n\tv_add_u32 %2, 0x12345, %2\n\tv_add_u32 %3, 0x12345, %3\n\tv_add_u32 %4, 0x12345, %4\n\tv_add_u32 %5, 0x12345, %5\n\tv_add_u32 %6, 0x12345, %6\n\tv_add_u32 %7, 0x12345, %7\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k40(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_and_b32 %0, 0x78787878, %0\n\tv_and_b32 %1, 0x78787878, %1\n\tv_and_b32 %2, 0x78787878, %2\n\tv_and_b32 %3, 0x78787878, %3\n\tv_and_b32 %4, 0x78787878, %4\n\tv_and_b32 %5, 0x78787878, %5\n\tv_and_b32 %6, 0x78787878, %6\n\tv_and_b32 %7, 0x78787878, %7\n\tv_and_b32 %0, 0x78787878, %0\n\tv_and_b32 %1, 0x78787878, %1\n\tv_and_b32 %2, 0x78787878, %2\n\tv_and_b32 %3, 0x78787878, %3\n\tv_and_b32 %4, 0x78787878, %4\n\tv_and_b32 %5, 0x78787878, %5\n\tv_and_b32 %6, 0x78787878, %6\n\tv_and_b32 %7, 0x78787878, %7\n\tv_and_b32 %0, 0x78787878, %0\n\tv_and_b32 %1, 0x78787878, %1\n\tv_and_b32 %2, 0x78787878, %2\n\tv_and_b32 %3, 0x78787878, %3\n\tv_and_b32 %4, 0x78787878, %4\n\tv_and_b32 %5, 0x78787878, %5\n\tv_and_b32 %6, 0x78787878, %6\n\tv_and_b32 %7, 0x78787878, %7\n\tv_and_b32 %0, 0x78787878, %0\n\tv_and_b32 %1, 0x78787878, %1\n\tv_and_b32 %2, 0x78787878, %2\n\tv_and_b32 %3, 0x78787878, %3\n\tv_and_b32 %4, 0x78787878, %4\n\tv_and_b32 %5, 0x78787878, %5\n\tv_and_b32 %6, 0x78787878, %6\n\tv_and_b32 %7, 0x78787878, %7\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k41(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_add_u32 %0, s4, %0\n\tv_add_u32 %1, s4, %1\n\tv_add_u32 %2, s4, %2\n\tv_add_u32 %3, s4, %3\n\tv_add_u32 %4, s4, %4\n\tv_add_u32 %5, s4, %5\n\tv_add_u32 %6, s4, %6\n\tv_add_u32 %7, s4, %7\n\tv_add_u32 %0, s4, %0\n\tv_add_u32 %1, s4, %1\n\tv_add_u32 %2, s4, %2\n\tv_add_u32 %3, s4, %3\n\tv_add_u32 %4, s4, %4\n\tv_add_u32 %5, s4, %5\n\tv_add_u32 %6, s4, %6\n\tv_add_u32 %7, s4, %7\n\tv_add_u32 %0, s4, %0\n\tv_add_u32 %1, s4, %1\n\tv_add_u32 %2, s4, %2\n\tv_add_u32 %3, s4, %3\n\tv_add_u32 %4, s4, %4\n\tv_add_u32 %5, s4, %5\n\tv_add_u32 %6, s4, %6\n\tv_add_u32 %7, s4, %7\n\tv_add_u32 %0, s4, %0\n\tv_add_u32 %1, s4, %1\n\tv_add_u32 %2, s4, %2\n\tv_add_u32 %3, s4, %3\n\tv_add_u32 %4, s4, %4\n\tv_add_u32 %5, s4, %5\n\tv_add_u32 %6, s4, %6\n\tv_add_u32 %7, s4, %7\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k42(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_subrev_u32 %0, %8, %0\n\tv_subrev_u32 %1, %8, %1\n\tv_subrev_u32 %2, %8, %2\n\tv_subrev_u32 %3, %8, %3\n\tv_subrev_u32 %4, %8, %4\n\tv_subrev_u32 %5, %8, %5\n\tv_subrev_u32 %6, %8, %6\n\tv_subrev_u32 %7, %8, %7\n\tv_subrev_u32 %0, %8, %0\n\tv_subrev_u32 %1, %8, %1\n\tv_subrev_u32 %2, %8, %2\n\tv_subrev_u32 %3, %8, %3\n\tv_subrev_u32 %4, %8, %4\n\tv_subrev_u32 %5, %8, %5\n\tv_subrev_u32 %6, %8, %6\n\tv_subrev_u32 %7, %8, %7\n\tv_subrev_u32 %0, %8, %0\n\tv_subrev_u32 %1, %8, %1\n\tv_subrev_u32 %2, %8, %2\n\tv_subrev_u32 %3, %8, %3\n\tv_subrev_u32 %4, %8, %4\n\tv_subrev_u32 %5, %8, %5\n\tv_subrev_u32 %6, %8, %6\n\tv_subrev_u32 %7, %8, %7\n\tv_subrev_u32 %0, %8, %0\n\tv_subrev_u32 %1, %8, %1\n\tv_subrev_u32 %2, %8, %2\n\tv_subrev_u32 %3, %8, %3\n\tv_subrev_u32 %4, %8, %4\n\tv_subrev_u32 %5, %8, %5\n\tv_subrev_u32 %6, %8, %6\n\tv_subrev_u32 %7, %8, %7\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k43(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_max_i32 %0, %0, %8\n\tv_max_i32 %1, %1, %8\n\tv_max_i32 %2, %2, %8\n\tv_max_i32 %3, %3, %8\n\tv_max_i32 %4, %4, %8\n\tv_max_i32 %5, %5, %8\n\tv_max_i32 %6, %6, %8\n\tv_max_i32 %7, %7, %8\n\tv_max_i32 %0, %0, %8\n\tv_max_i32 %1, %1, %8\n\tv_max_i32 %2, %2, %8\n\tv_max_i32 %3, %3, %8\n\tv_max_i32 %4, %4, %8\n\tv_max_i32 %5, %5, %8\n\tv_max_i32 %6, %6, %8\n\tv_max_i32 %7, %7, %8\n\tv_max_i32 %0, %0, %8\n\tv_max_i32 %1, %1, %8\n\tv_max_i32 %2, %2, %8\n\tv_max_i32 %3, %3, %8\n\tv_max_i32 %4, %4, %8\n\tv_max_i32 %5, %5, %8\n\tv_max_i32 %6, %6, %8\n\tv_max_i32 %7, %7, %8\n\tv_max_i32 %0, %0, %8\n\tv_max_i32 %1, %1, %8\n\tv_max_i32 %2, %2, %8\n\tv_max_i32 %3, %3, %8\n\tv_max_i32 %4, %4, %8\n\tv_max_i32 %5, %5, %8\n\tv_max_i32 %6, %6, %8\n\tv_max_i32 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k44(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_cndmask_b32 %0, %8, %8, vcc\n\tv_cndmask_b32 %1, %8, %8, vcc\n\tv_cndmask_b32 %2, %8, %8, vcc\n\tv_cndmask_b32 %3, %8, %8, vcc\n\tv_cndmask_b32 %4, %8, %8, vcc\n\tv_cndmask_b32 %5, %8, %8, vcc\n\tv_cndmask_b32 %6, %8, %8, vcc\n\tv_cndmask_b32 %7, %8, %8, vcc\n\tv_cndmask_b32 %0, %8, %8, vcc\n\tv_cndmask_b32 %1, %8, %8, vcc\n\tv_cndmask_b32 %2, %8, %8, vcc\n\tv_cndmask_b32 %3, %8, %8, vcc\n\tv_cndmask_b32 %4, %8, %8, vcc\n\tv_cndmask_b32 %5, %8, %8, vcc\n\tv_cndmask_b32 %6, %8, %8, vcc\n\tv_cndmask_b32 %7, %8, %8, vcc\n\tv_cndmask_b32 %0, %8, %8, vcc\n\tv_cndmask_b32 %1, %8, %8, vcc\n\tv_cndmask_b32 %2, %8, %8, vcc\n\tv_cndmask_b32 %3, %8, %8, vcc\n\tv_cndmask_b32 %4, %8, %8, vcc\n\tv_cndmask_b32 %5, %8, %8, vcc\n\tv_cndmask_b32 %6, %8, %8, vcc\n\tv_cndmask_b32 %7, %8, %8, vcc\n\tv_cndmask_b32 %0, %8, %8, vcc\n\tv_cndmask_b32 %1, %8, %8, vcc\n\tv_cndmask_b32 %2, %8, %8, vcc\n\tv_cndmask_b32 %3, %8, %8, vcc\n\tv_cndmask_b32 %4, %8, %8, vcc\n\tv_cndmask_b32 %5, %8, %8, vcc\n\tv_cndmask_b32 %6, %8, %8, vcc\n\tv_cndmask_b32 %7, %8, %8, vcc\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k45(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_bitop3_b32 %0, %0, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %1, %1, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %2, %2, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %3, %3, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %4, %4, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %5, %5, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %6, %6, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %7, %7, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %0, %0, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %1, %1, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %2, %2, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %3, %3, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %4, %4, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %5, %5, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %6, %6, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %7, %7, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %0, %0, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %1, %1, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %2, %2, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %3, %3, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %4, %4, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %5, %5, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %6, %6, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %7, %7, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %0, %0, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %1, %1, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %2, %2, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %3, %3, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %4, %4, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %5, %5, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %6, %6, %8, %8 bitop3:0xb\n\tv_bitop3_b32 %7, %7, %8, %8 bitop3:0xb\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k46(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_xnor_b32 %0, %0, %8\n\tv_xnor_b32 %1, %1, %8\n\tv_xnor_b32 %2, %2, %8\n\tv_xnor_b32 %3, %3, %8\n\tv_xnor_b32 %4, %4, %8\n\tv_xnor_b32 %5, %5, %8\n\tv_xnor_b32 %6, %6, %8\n\tv_xnor_b32 %7, %7, %8\n\tv_xnor_b32 %0, %0, %8\n\tv_xnor_b32 %1, %1, %8\n\tv_xnor_b32 %2, %2, %8\n\tv_xnor_b32 %3, %3, %8\n\tv_xnor_b32 %4, %4, %8\n\tv_xnor_b32 %5, %5, %8\n\tv_xnor_b32 %6, %6, %8\n\tv_xnor_b32 %7, %7, %8\n\tv_xnor_b32 %0, %0, %8\n\tv_xnor_b32 %1, %1, %8\n\tv_xnor_b32 %2, %2, %8\n\tv_xnor_b32 %3, %3, %8\n\tv_xnor_b32 %4, %4, %8\n\tv_xnor_b32 %5, %5, %8\n\tv_xnor_b32 %6, %6, %8\n\tv_xnor_b32 %7, %7, %8\n\tv_xnor_b32 %0, %0, %8\n\tv_xnor_b32 %1, %1, %8\n\tv_xnor_b32 %2, %2, %8\n\tv_xnor_b32 %3, %3, %8\n\tv_xnor_b32 %4, %4, %8\n\tv_xnor_b32 %5, %5, %8\n\tv_xnor_b32 %6, %6, %8\n\tv_xnor_b32 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k47(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_sad_u32 %0, %0, %8, %8\n\tv_sad_u32 %1, %1, %8, %8\n\tv_sad_u32 %2, %2, %8, %8\n\tv_sad_u32 %3, %3, %8, %8\n\tv_sad_u32 %4, %4, %8, %8\n\tv_sad_u32 %5, %5, %8, %8\n\tv_sad_u32 %6, %6, %8, %8\n\tv_sad_u32 %7, %7, %8, %8\n\tv_sad_u32 %0, %0, %8, %8\n\tv_sad_u32 %1, %1, %8, %8\n\tv_sad_u32 %2, %2, %8, %8\n\tv_sad_u32 %3, %3, %8, %8\n\tv_sad_u32 %4, %4, %8, %8\n\tv_sad_u32 %5, %5, %8, %8\n\tv_sad_u32 %6, %6, %8, %8\n\tv_sad_u32 %7, %7, %8, %8\n\tv_sad_u32 %0, %0, %8, %8\n\tv_sad_u32 %1, %1, %8, %8\n\tv_sad_u32 %2, %2, %8, %8\n\tv_sad_u32 %3, %3, %8, %8\n\tv_sad_u32 %4, %4, %8, %8\n\tv_sad_u32 %5, %5, %8, %8\n\tv_sad_u32 %6, %6, %8, %8\n\tv_sad_u32 %7, %7, %8, %8\n\tv_sad_u32 %0, %0, %8, %8\n\tv_sad_u32 %1, %1, %8, %8\n\tv_sad_u32 %2, %2, %8, %8\n\tv_sad_u32 %3, %3, %8, %8\n\tv_sad_u32 %4, %4, %8, %8\n\tv_sad_u32 %5, %5, %8, %8\n\tv_sad_u32 %6, %6, %8, %8\n\tv_sad_u32 %7, %7, %8, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k48(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_add_lshl_u32 %0, %0, %8, 10\n\tv_add_lshl_u32 %1, %1, %8, 10\n\tv_add_lshl_u32 %2, %2, %8, 10\n\tv_add_lshl_u32 %3, %3, %8, 10\n\tv_add_lshl_u32 %4, %4, %8, 10\n\tv_add_lshl_u32 %5, %5, %8, 10\n\tv_add_lshl_u32 %6, %6, %8, 10\n\tv_add_lshl_u32 %7, %7, %8, 10\n\tv_add_lshl_u32 %0, %0, %8, 10\n\tv_add_lshl_u32 %1, %1, %8, 10\n\tv_add_lshl_u32 %2, %2, %8, 10\n\tv_add_lshl_u32 %3, %3, %8, 10\n\tv_add_lshl_u32 %4, %4, %8, 10\n\tv_add_lshl_u32 %5, %5, %8, 10\n\tv_add_lshl_u32 %6, %6, %8, 10\n\tv_add_lshl_u32 %7, %7, %8, 10\n\tv_add_lshl_u32 %0, %0, %8, 10\n\tv_add_lshl_u32 %1, %1, %8, 10\n\tv_add_lshl_u32 %2, %2, %8, 10\n\tv_add_lshl_u32 %3, %3, %8, 10\n\tv_add_lshl_u32 %4, %4, %8, 10\n\tv_add_lshl_u32 %5, %5, %8, 10\n\tv_add_lshl_u32 %6, %6, %8, 10\n\tv_add_lshl_u32 %7, %7, %8, 10\n\tv_add_lshl_u32 %0, %0, %8, 10\n\tv_add_lshl_u32 %1, %1, %8, 10\n\tv_add_lshl_u32 %2, %2, %8, 10\n\tv_add_lshl_u32 %3, %3, %8, 10\n\tv_add_lshl_u32 %4, %4, %8, 10\n\tv_add_lshl_u32 %5, %5, %8, 10\n\tv_add_lshl_u32 %6, %6, %8, 10\n\tv_add_lshl_u32 %7, %7, %8, 10\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k49(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_mbcnt_lo_u32_b32 %0, %0, %8\n\tv_mbcnt_lo_u32_b32 %1, %1, %8\n\tv_mbcnt_lo_u32_b32 %2, %2, %8\n\tv_mbcnt_lo_u32_b32 %3, %3, %8\n\tv_mbcnt_lo_u32_b32 %4, %4, %8\n\tv_mbcnt_lo_u32_b32 %5, %5, %8\n\tv_mbcnt_lo_u32_b32 %6, %6, %8\n\tv_mbcnt_lo_u32_b32 %7, %7, %8\n\tv_mbcnt_lo_u32_b32 %0, %0, %8\n\tv_mbcnt_lo_u32_b32 %1, %1, %8\n\tv_mbcnt_lo_u32_b32 %2, %2, %8\n\tv_mbcnt_lo_u32_b32 %3, %3, %8\n\tv_mbcnt_lo_u32_b32 %4, %4, %8\n\tv_mbcnt_lo_u32_b32 %5, %5, %8\n\tv_mbcnt_lo_u32_b32 %6, %6, %8\n\tv_mbcnt_lo_u32_b32 %7, %7, %8\n\tv_mbcnt_lo_u32_b32 %0, %0, %8\n\tv_mbcnt_lo_u32_b32 %1, %1, %8\n\tv_mbcnt_lo_u32_b32 %2, %2, %8\n\tv_mbcnt_lo_u32_b32 %3, %3, %8\n\tv_mbcnt_lo_u32_b32 %4, %4, %8\n\tv_mbcnt_lo_u32_b32 %5, %5, %8\n\tv_mbcnt_lo_u32_b32 %6, %6, %8\n\tv_mbcnt_lo_u32_b32 %7, %7, %8\n\tv_mbcnt_lo_u32_b32 %0, %0, %8\n\tv_mbcnt_lo_u32_b32 %1, %1, %8\n\tv_mbcnt_lo_u32_b32 %2, %2, %8\n\tv_mbcnt_lo_u32_b32 %3, %3, %8\n\tv_mbcnt_lo_u32_b32 %4, %4, %8\n\tv_mbcnt_lo_u32_b32 %5, %5, %8\n\tv_mbcnt_lo_u32_b32 %6, %6, %8\n\tv_mbcnt_lo_u32_b32 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k50(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_bfi_b32 %0, %0, %8, %8\n\tv_bfi_b32 %1, %1, %8, %8\n\tv_bfi_b32 %2, %2, %8, %8\n\tv_bfi_b32 %3, %3, %8, %8\n\tv_bfi_b32 %4, %4, %8, %8\n\tv_bfi_b32 %5, %5, %8, %8\n\tv_bfi_b32 %6, %6, %8, %8\n\tv_bfi_b32 %7, %7, %8, %8\n\tv_bfi_b32 %0, %0, %8, %8\n\tv_bfi_b32 %1, %1, %8, %8\n\tv_bfi_b32 %2, %2, %8, %8\n\tv_bfi_b32 %3, %3, %8, %8\n\tv_bfi_b32 %4, %4, %8, %8\n\tv_bfi_b32 %5, %5, %8, %8\n\tv_bfi_b32 %6, %6, %8, %8\n\tv_bfi_b32 %7, %7, %8, %8\n\tv_bfi_b32 %0, %0, %8, %8\n\tv_bfi_b32 %1, %1, %8, %8\n\tv_bfi_b32 %2, %2, %8, %8\n\tv_bfi_b32 %3, %3, %8, %8\n\tv_bfi_b32 %4, %4, %8, %8\n\tv_bfi_b32 %5, %5, %8, %8\n\tv_bfi_b32 %6, %6, %8, %8\n\tv_bfi_b32 %7, %7, %8, %8\n\tv_bfi_b32 %0, %0, %8, %8\n\tv_bfi_b32 %1, %1, %8, %8\n\tv_bfi_b32 %2, %2, %8, %8\n\tv_bfi_b32 %3, %3, %8, %8\n\tv_bfi_b32 %4, %4, %8, %8\n\tv_bfi_b32 %5, %5, %8, %8\n\tv_bfi_b32 %6, %6, %8, %8\n\tv_bfi_b32 %7, %7, %8, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k51(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_min_u16 %0, %0, %8\n\tv_min_u16 %1, %1, %8\n\tv_min_u16 %2, %2, %8\n\tv_min_u16 %3, %3, %8\n\tv_min_u16 %4, %4, %8\n\tv_min_u16 %5, %5, %8\n\tv_min_u16 %6, %6, %8\n\tv_min_u16 %7, %7, %8\n\tv_min_u16 %0, %0, %8\n\tv_min_u16 %1, %1, %8\n\tv_min_u16 %2, %2, %8\n\tv_min_u16 %3, %3, %8\n\tv_min_u16 %4, %4, %8\n\tv_min_u16 %5, %5, %8\n\tv_min_u16 %6, %6, %8\n\tv_min_u16 %7, %7, %8\n\tv_min_u16 %0, %0, %8\n\tv_min_u16 %1, %1, %8\n\tv_min_u16 %2, %2, %8\n\tv_min_u16 %3, %3, %8\n\tv_min_u16 %4, %4, %8\n\tv_min_u16 %5, %5, %8\n\tv_min_u16 %6, %6, %8\n\tv_min_u16 %7, %7, %8\n\tv_min_u16 %0, %0, %8\n\tv_min_u16 %1, %1, %8\n\tv_min_u16 %2, %2, %8\n\tv_min_u16 %3, %3, %8\n\tv_min_u16 %4, %4, %8\n\tv_min_u16 %5, %5, %8\n\tv_min_u16 %6, %6, %8\n\tv_min_u16 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k52(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_pk_min_u16 %0, %0, %8\n\tv_pk_min_u16 %1, %1, %8\n\tv_pk_min_u16 %2, %2, %8\n\tv_pk_min_u16 %3, %3, %8\n\tv_pk_min_u16 %4, %4, %8\n\tv_pk_min_u16 %5, %5, %8\n\tv_pk_min_u16 %6, %6, %8\n\tv_pk_min_u16 %7, %7, %8\n\tv_pk_min_u16 %0, %0, %8\n\tv_pk_min_u16 %1, %1, %8\n\tv_pk_min_u16 %2, %2, %8\n\tv_pk_min_u16 %3, %3, %8\n\tv_pk_min_u16 %4, %4, %8\n\tv_pk_min_u16 %5, %5, %8\n\tv_pk_min_u16 %6, %6, %8\n\tv_pk_min_u16 %7, %7, %8\n\tv_pk_min_u16 %0, %0, %8\n\tv_pk_min_u16 %1, %1, %8\n\tv_pk_min_u16 %2, %2, %8\n\tv_pk_min_u16 %3, %3, %8\n\tv_pk_min_u16 %4, %4, %8\n\tv_pk_min_u16 %5, %5, %8\n\tv_pk_min_u16 %6, %6, %8\n\tv_pk_min_u16 %7, %7, %8\n\tv_pk_min_u16 %0, %0, %8\n\tv_pk_min_u16 %1, %1, %8\n\tv_pk_min_u16 %2, %2, %8\n\tv_pk_min_u16 %3, %3, %8\n\tv_pk_min_u16 %4, %4, %8\n\tv_pk_min_u16 %5, %5, %8\n\tv_pk_min_u16 %6, %6, %8\n\tv_pk_min_u16 %7, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k53(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_sub_co_u32 %0, vcc, %0, %8\n\tv_sub_co_u32 %1, vcc, %1, %8\n\tv_sub_co_u32 %2, vcc, %2, %8\n\tv_sub_co_u32 %3, vcc, %3, %8\n\tv_sub_co_u32 %4, vcc, %4, %8\n\tv_sub_co_u32 %5, vcc, %5, %8\n\tv_sub_co_u32 %6, vcc, %6, %8\n\tv_sub_co_u32 %7, vcc, %7, %8\n\tv_sub_co_u32 %0, vcc, %0, %8\n\tv_sub_co_u32 %1, vcc, %1, %8\n\tv_sub_co_u32 %2, vcc, %2, %8\n\tv_sub_co_u32 %3, vcc, %3, %8\n\tv_sub_co_u32 %4, vcc, %4, %8\n\tv_sub_co_u32 %5, vcc, %5, %8\n\tv_sub_co_u32 %6, vcc, %6, %8\n\tv_sub_co_u32 %7, vcc, %7, %8\n\tv_sub_co_u32 %0, vcc, %0, %8\n\tv_sub_co_u32 %1, vcc, %1, %8\n\tv_sub_co_u32 %2, vcc, %2, %8\n\tv_sub_co_u32 %3, vcc, %3, %8\n\tv_sub_co_u32 %4, vcc, %4, %8\n\tv_sub_co_u32 %5, vcc, %5, %8\n\tv_sub_co_u32 %6, vcc, %6, %8\n\tv_sub_co_u32 %7, vcc, %7, %8\n\tv_sub_co_u32 %0, vcc, %0, %8\n\tv_sub_co_u32 %1, vcc, %1, %8\n\tv_sub_co_u32 %2, vcc, %2, %8\n\tv_sub_co_u32 %3, vcc, %3, %8\n\tv_sub_co_u32 %4, vcc, %4, %8\n\tv_sub_co_u32 %5, vcc, %5, %8\n\tv_sub_co_u32 %6, vcc, %6, %8\n\tv_sub_co_u32 %7, vcc, %7, %8\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s4");
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
typedef void (*kern)(uint32_t *, int, uint32_t);
static void run(const char *name, kern k, uint32_t *d) {
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
    const double ops = (double)blocks * 256 * (double)iters * 32;
    printf("%-16s %.3f ms  %6.2f T lane-ops/s\n", name, ms, ops / (ms * 1e-3) / 1e12);
}
int main() {
    uint32_t *d;
    (void)hipMalloc(&d, 8192 * 256 * 4);
    run("v_add_u32", k0, d);
    run("v_xor_b32", k1, d);
    run("v_and_b32", k2, d);
    run("v_or_b32", k3, d);
    run("v_min_u32", k4, d);
    run("v_max_u32", k5, d);
    run("v_min3_u32", k6, d);
    run("v_lshlrev_b32", k7, d);
    run("v_lshrrev_b32", k8, d);
    run("v_bitop3_b32", k9, d);
    run("v_alignbit_b32", k10, d);
    run("v_alignbit_v", k11, d);
    run("v_lshl_add_u32", k12, d);
    run("v_add3_u32", k13, d);
    run("v_lshl_or_b32", k14, d);
    run("v_and_or_b32", k15, d);
    run("v_bfe_u32", k16, d);
    run("v_bfe_u32_v", k17, d);
    run("v_mad_u32_u24", k18, d);
    run("v_mul_lo_u32", k19, d);
    run("v_bcnt_u32_b32", k20, d);
    run("v_add_co_u32", k21, d);
    run("v_addc_co_u32", k22, d);
    run("v_cndmask_b32", k23, d);
    run("v_cmp_lt_u32", k24, d);
    run("v_sub_u32", k25, d);
    run("v_xad_u32", k26, d);
    run("v_perm_b32", k27, d);
    run("v_mov_b32", k28, d);
    run("v_alignbyte_b32", k29, d);
    run("v_pk_add_u16", k30, d);
    run("v_ffbh_u32", k31, d);
    run("v_bfrev_b32", k32, d);
    run("v_lshlrev_b64", k33, d);
    run("v_ashrrev_i32", k34, d);
    run("v_lshlrev_b32_v", k35, d);
    run("v_lshrrev_b32_v", k36, d);
    run("v_not_b32", k37, d);
    run("v_add_u32_self", k38, d);
    run("v_add_u32_lit", k39, d);
    run("v_and_b32_lit", k40, d);
    run("v_add_u32_sgpr", k41, d);
    run("v_subrev_u32", k42, d);
    run("v_max_i32", k43, d);
    run("v_cndmask_nochain", k44, d);
    run("v_bitop3_2in", k45, d);
    run("v_xnor_b32", k46, d);
    run("v_sad_u32", k47, d);
    run("v_add_lshl_u32", k48, d);
    run("v_mbcnt_lo", k49, d);
    run("v_bfi_b32", k50, d);
    run("v_min_u16", k51, d);
    run("v_pk_min_u16", k52, d);
    run("v_sub_co_u32", k53, d);
    return 0;
}
