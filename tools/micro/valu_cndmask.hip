// valu_cndmask.hip -- v_cndmask_b32 issue rate in a few settings (the plain table shows 7.6 T).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define R8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define C_VCC(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define C_SG(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[6:7]\n\t"
#define C_ADD(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\tv_add_u32 %" #i ", %" #i ", %8\n\t"

template <int KIND>
__global__ __launch_bounds__(256) void spin(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1u;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0)        // vcc written by a VALU compare right before
            asm volatile("v_cmp_lt_u32 vcc, %0, %8\n\ts_nop 4\n\t" R8(C_VCC) R8(C_VCC) R8(C_VCC) R8(C_VCC)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
        else if (KIND == 1)   // condition in an SGPR pair
            asm volatile("v_cmp_lt_u32 s[6:7], %0, %8\n\ts_nop 4\n\t" R8(C_SG) R8(C_SG) R8(C_SG) R8(C_SG)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s6", "s7");
        else if (KIND == 2)   // vcc = all ones (scalar write)
            asm volatile("s_mov_b64 vcc, -1\n\ts_nop 4\n\t" R8(C_VCC) R8(C_VCC) R8(C_VCC) R8(C_VCC)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
        else                  // cndmask alternating with add (16 + 16 ops)
            asm volatile("v_cmp_lt_u32 vcc, %0, %8\n\ts_nop 4\n\t" R8(C_ADD) R8(C_ADD)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int KIND>
static void run(const char *name, uint32_t *d) {
    const int blocks = 8192, iters = 1024;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(spin<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.3f ms  %.2f SIMD-cycles per VALU instruction (2.4 GHz)\n", name, ms,
           ms * 1e-3 * 2.4e9 * 1024 / ((double)blocks * 4 * iters * 33));
}

int main() {
    uint32_t *d;
    (void)hipMalloc(&d, 8192 * 256 * 4);
    run<0>("cndmask vcc (v_cmp)", d);
    run<1>("cndmask s[6:7] (v_cmp)", d);
    run<2>("cndmask vcc (s_mov -1)", d);
    run<3>("cndmask + add", d);
    return 0;
}
